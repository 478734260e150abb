"""Time phx_step of SC64 at one batch size with one step-kernel variant, replayed from a hipGraph of 50 launches.
   python tools/step_variant_time.py <batch> <auto|fused|wide>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, phantom_amd as ph
B=int(sys.argv[1]); v=sys.argv[2]
env = ph.SupplyChainEnv(n_shops=9, customers_per_shop=6, num_steps=100, batch_size=B, seed=42, exogenous="device", variants={"step": v})
env.reset(); dev=env._device()
acts=(torch.rand(50,B,9,device="cuda")*100).contiguous()
for t in range(10): dev.step(acts[t])
sg=dev.step_graph(acts); sg.replay(); torch.cuda.synchronize()
best=1e9
for r in range(3):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): sg.replay()
    e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/200*1e3)
print(f"B={B:7d} {v:6s} {best:7.2f} us/step (hipGraph)  {dev.last_kernel()}")
