// phx_sc_policy.hip -- a fused supply-chain rollout whose POLICY is evaluated on the device.
//
// The reference's collection loop calls a policy for every agent and step (utils/rllib/rollout.py:300-363: compute_action on the
// agent's last observation, then env.step).  With the policy outside the library that is one launch per step with the policy's
// kernels between the launches (bench.py `on_policy`: 34.7 us per step at SC64, B = 4096, of which 22 a 3-32-1 torch MLP); the
// store-wave kernel cannot serve it either: it draws its actions two chunks AHEAD of the stock chain, and a policy's action
// depends on the observation the previous step produced.  This kernel is sequential in time where the problem is:
//   * one lane per (env, shop) pair walks T steps: MLP on the previous observation (phx_policy_mlp: the arithmetic is defined
//     in include/phantom_amd.h, fmaf term by term, and restated by the oracle), decode_action, the customers' orders (device
//     Philox stream -- one block serves four ticks -- or replayed draws), the closed form of the step's message exchange
//     (supply_chain.py:98-122,136-142: stock' = stock - min(stock, D) + min(R, 100 - stock)), encode_observation /
//     compute_reward from registers (IEEE f32 divisions, the f64 reward rounded once), the trajectory row, auto-reset;
//   * the weights are the same for every lane: they are staged in LDS once per workgroup (at most 17.7 KB) and read as broadcast
//     16-byte pieces, four weights per instruction (the first form read them through the scalar cache as SGPR operands: scalar
//     loads return out of order, every use waited for all of them -- 2.6 us per step for a 3-32-1 network, 161 for 3-64-64-1);
//     a second hidden layer keeps the first one's activations in a wave-private LDS column per lane ([unit][lane]: bank-conflict
//     free) because a register array cannot be indexed by a run-time width;
//   * a workgroup holds WHOLE envs (floor(256 / S) of them; 128 lanes with two hidden layers): an env's step counter and tick are read and written by one
//     workgroup only.
// Bound: at B = 4096 the launch is 0.6 waves per SIMD and every step a dependent chain (~100 fmas + the output accumulation)
// -- latency, not HBM: 22 S bytes per env-step leave as plain per-lane stores.
#include "phx_dev.h"

#include <cstring>

struct PolArgs {
  int32_t B, S, epb, T, num_steps, n_exo;
  uint64_t seed; int64_t env_offset;
  int32_t *stock, *sales, *missed, *delivered, *env_step, *env_tick;
  const int32_t* shop_norm; const int32_t* shop_cust_ptr; const int32_t* shop_cust_exo;
  phx_rollout_io io;
  phx_policy_mlp pol;
};

// act(c) as ONE v_med3_f32: ReLU = the median of (c, 0, +inf), hard-tanh = the median of (c, -1, 1).  Finite c: the values of the header's
// definition (c > 0 ? c : +0 / the two-sided clip); a -0 that med3 may let through where the definition says +0 changes no sum's value and the
// definition's closing "+ 0.0f" removes it from the action.
template <int ACT>
__device__ __forceinline__ float pol_act(float c) {
  if (ACT == PHX_ACT_HARD_TANH) return __builtin_amdgcn_fmed3f(c, -1.0f, 1.0f);
  return __builtin_amdgcn_fmed3f(c, 0.0f, __builtin_inff());
}
typedef float pol_f2 __attribute__((ext_vector_type(2)));

// LDS image of the network (floats), staged once per workgroup -- every lane reads the same addresses (broadcast reads, 16 bytes =
// four weights per instruction; LDS returns in order, so the compiler keeps many reads in flight, which scalar loads do not allow).
// Hidden widths are PADDED to multiples of 8 with zero weights and zero biases: a padded unit is act(0) = 0 and adds fmaf(0, 0, c) = c
// to every sum it enters -- the value of every sum is unchanged (only the sign of an exact zero can differ, which the definition's
// closing "+ 0.0f" removes from the action), and no loop below carries a guard.
//   [0, 4 W0p)                      layer 0, per PAIR of units (2p, 2p + 1): (w[.][0] x 2, w[.][1] x 2 | w[.][2] x 2, b[.] x 2) -- two units per v_pk_fma_f32
//   one hidden layer:  o1 = 4 W0p:  the output row w[1][0 .. W0p), then b[1][0] (+ 3 pad)
//   two hidden layers: o1 = 4 W0p:  layer 1, W1p rows of W0p floats; then b[1][0 .. W1p), the output row w[2][0 .. W1p), b[2][0] (+ 3 pad)
__host__ __device__ inline int pol_img_floats(int n_hidden, int W0p, int W1p) {
  return n_hidden == 1 ? 4 * W0p + W0p + 4 : 4 * W0p + W1p * W0p + W1p + W1p + 4;
}

// NT threads per workgroup: 256 with one hidden layer; 128 with two (the first layer's activations live in a wave-private LDS column
// per lane, [unit][lane]: bank-conflict free -- a register array cannot be indexed by a run-time width)
// EXO: the order sizes are replayed from io.exo (global loads inside the step loop); the device-drawn instantiation has NO load in its loop, so
// that no s_waitcnt vmcnt ever waits for the previous step's trajectory stores (loads and stores share the counter on gfx950)
// SW (two hidden layers whose widths are multiples of 8 and 4): the second layer's weights as SGPR operands -- four rows x eight k per trip
// through the scalar cache (s_load_dwordx8 from torch's own [W1][W0] layout), 32 fused multiply-adds on four independent chains; LDS then
// carries the first layer's activations only (one 4-byte read per lane and k, shared by the four rows).  The LDS form reads every weight as
// a broadcast: 64 lanes x 4 bytes of LDS return bandwidth per multiply-add -- the layer was LDS-bound (32 us per step for 3-64-64-1).
template <int ACT, bool TWO, int NT, bool EXO, bool SW>
__global__ __launch_bounds__(NT) void phx_sc_rollout_policy_kernel(const PolArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_img[];
  const int tid = threadIdx.x, S = a.S;
  const int W0 = a.pol.width[0], W1 = TWO ? a.pol.width[1] : 0;
  const int W0p = (W0 + 7) & ~7, W1p = (W1 + 7) & ~7;
  const int n_img = pol_img_floats(TWO ? 2 : 1, W0p, W1p);
  float* const s_h = s_img + n_img;                                    // TWO: [W0p][NT]
  {                                                                    // stage the network
    for (int i = tid; i < W0p; i += NT) {
      const bool in = i < W0;
      float* const q = s_img + 8 * (i >> 1) + (i & 1);
      q[0] = in ? a.pol.w[0][i * 3 + 0] : 0.0f; q[2] = in ? a.pol.w[0][i * 3 + 1] : 0.0f;
      q[4] = in ? a.pol.w[0][i * 3 + 2] : 0.0f; q[6] = in ? a.pol.b[0][i] : 0.0f;
    }
    float* o1 = s_img + 4 * W0p;
    if (!TWO) {
      for (int i = tid; i < W0p + 4; i += NT) o1[i] = i < W0 ? a.pol.w[1][i] : (i == W0p ? a.pol.b[1][0] : 0.0f);
    } else {
      for (int i = tid; i < W1p * W0p; i += NT) { const int r = i / W0p, k = i - r * W0p; o1[i] = (r < W1 && k < W0) ? a.pol.w[1][r * W0 + k] : 0.0f; }
      float* ob1 = o1 + W1p * W0p; float* ow2 = ob1 + W1p;
      for (int i = tid; i < W1p; i += NT) { ob1[i] = i < W1 ? a.pol.b[1][i] : 0.0f; ow2[i] = i < W1 ? a.pol.w[2][i] : 0.0f; }
      if (tid < 4) ow2[W1p + tid] = tid == 0 ? a.pol.b[2][0] : 0.0f;
    }
  }
  const int b0 = (int)blockIdx.x * a.epb;                              // the workgroup's first env
  const int n_env = min(a.epb, a.B - b0);
  const bool on = tid < n_env * S;
  const int el = on ? tid / S : 0, s = on ? tid - el * S : 0;
  const int b = b0 + el;
  const int64_t pair = (int64_t)b * S + s, total = (int64_t)a.B * S;
  int stock = a.stock[pair], sales = a.sales[pair], missed = a.missed[pair], delivered = a.delivered[pair];
  int step = a.env_step[b]; uint32_t tick = (uint32_t)a.env_tick[b];
  // (a "use" of every loaded word HERE: `delivered` is overwritten by the first step without ever being read, and the s_waitcnt vmcnt(0) that
  //  protects its register from the load still in flight would otherwise sit inside the step loop -- where it waits for the row's stores)
  asm volatile("" :: "v"(stock), "v"(sales), "v"(missed), "v"(delivered), "v"(step), "v"(tick));
  const int norm_i = a.shop_norm[s];
  const float norm_f = (float)norm_i;
  const int c0 = a.shop_cust_ptr[s], K = a.shop_cust_ptr[s + 1] - c0;
  const int64_t genv = a.env_offset + b;
  const float out_scale = a.pol.out_scale, out_bias = a.pol.out_bias, out_lo = a.pol.out_lo, out_hi = a.pol.out_hi;
  // the divisors never change: their reciprocals once (IEEE divisions), a quotient = a multiply and Markstein's correction (phx_dev.h: div_by_recip)
  const float r_stock = 1.0f / (float)PHX_SHOP_MAX_STOCK, r_norm = 1.0f / norm_f;
  const bool norm_small = norm_i >= 1 && norm_i <= DIV_RECIP_N;
  auto encode = [&](int st, int sl, int ms, float* o) {                // ShopAgent.encode_observation, supply_chain.py:124-134
    if (__builtin_expect(norm_small && (unsigned)(st | sl | ms) < (unsigned)DIV_RECIP_X, 1)) {
      o[0] = div_by_recip((float)st, (float)PHX_SHOP_MAX_STOCK, r_stock);
      o[1] = div_by_recip((float)sl, norm_f, r_norm);
      o[2] = div_by_recip((float)ms, norm_f, r_norm);
    } else if ((((unsigned)st + (1u << 24)) | ((unsigned)sl + (1u << 24)) | ((unsigned)ms + (1u << 24)) | ((unsigned)norm_i + (1u << 24))) < (2u << 24))
      shop_obs_f32(st, sl, ms, norm_f, o);
    else shop_obs(st, sl, ms, norm_i, o);
  };
  float x[3];
  encode(stock, sales, missed, x);                                     // what the agent observes now: the policy's first input
  const bool small_k = __all(K <= 6) != 0;
  const uint32_t pK = K <= 0 ? 1u : K == 1 ? 5u : K == 2 ? 25u : K == 3 ? 125u : K == 4 ? 625u : K == 5 ? 3125u : 15625u;
  const float inv_pK = K <= 0 ? 1.0f : K == 1 ? 0.2f : K == 2 ? 0.04f : K == 3 ? 0.008f : K == 4 ? 0.0016f : K == 5 ? 0.00032f : 0.000064f;
  RngQuadCache quad; quad.q = 0xffffffffu; quad.w[0] = quad.w[1] = quad.w[2] = quad.w[3] = 0u;
  float* const hcol = s_h + tid;
  const float4* const img0 = (const float4*)s_img;                     // layer 0, one float4 per unit
  const float* const o1 = s_img + 4 * W0p;
  __syncthreads();

  float* p_obs = a.io.obs + pair * 3; float* p_act = a.io.action_out + pair; float* p_rew = a.io.reward + pair;
  const bool has_ter = a.io.terminated != nullptr;                     // (uniform: a scalar branch)
  uint8_t* p_ter = a.io.terminated + pair; uint8_t* p_tru = a.io.truncated + pair;
  for (int t = 0; t < a.T; ++t) {
    // ---- compute_action: the MLP on the previous observation (phx_policy_mlp, include/phantom_amd.h) --------------------------------
    // two units per packed fused multiply-add (each element is the fmaf of the definition)
    auto pair0 = [&](int p, float& ha, float& hb) __attribute__((always_inline)) {
      const float4 u = img0[2 * p], v = img0[2 * p + 1];
      pol_f2 c = {v.z, v.w};
      c = __builtin_elementwise_fma((pol_f2){u.x, u.y}, (pol_f2){x[0], x[0]}, c);
      c = __builtin_elementwise_fma((pol_f2){u.z, u.w}, (pol_f2){x[1], x[1]}, c);
      c = __builtin_elementwise_fma((pol_f2){v.x, v.y}, (pol_f2){x[2], x[2]}, c);
      ha = pol_act<ACT>(c.x); hb = pol_act<ACT>(c.y);
    };
    float y;
    if (!TWO) {
      y = o1[W0p];
      for (int i0 = 0; i0 < W0p; i0 += 8) {                            // eight units at a time: their reads are in flight together
        float h[8];
#pragma unroll
        for (int u = 0; u < 8; u += 2) pair0((i0 + u) >> 1, h[u], h[u + 1]);
        const float4 wa = *(const float4*)(o1 + i0), wb = *(const float4*)(o1 + i0 + 4);
        y = __fmaf_rn(wa.x, h[0], y); y = __fmaf_rn(wa.y, h[1], y); y = __fmaf_rn(wa.z, h[2], y); y = __fmaf_rn(wa.w, h[3], y);      // ascending unit order
        y = __fmaf_rn(wb.x, h[4], y); y = __fmaf_rn(wb.y, h[5], y); y = __fmaf_rn(wb.z, h[6], y); y = __fmaf_rn(wb.w, h[7], y);
      }
    } else {
      for (int i0 = 0; i0 < W0p; i0 += 8) {
        float h[8];
#pragma unroll
        for (int u = 0; u < 8; u += 2) pair0((i0 + u) >> 1, h[u], h[u + 1]);
#pragma unroll
        for (int u = 0; u < 8; ++u) hcol[(i0 + u) * NT] = h[u];
      }
      const float* const ob1 = o1 + W1p * W0p; const float* const ow2 = ob1 + W1p;
      y = ow2[W1p];
      if (SW && ((W0 | W1) & 7) == 0) {
        // eight units x eight k per trip: 64 weights in SGPRs (eight s_load_dwordx8 in flight together), 64 v_fmac_f32 with an SGPR operand on
        // eight independent chains (ascending k each) -- no packing moves, one scalar-cache and one LDS round trip per 64 multiply-adds
        typedef const __attribute__((address_space(4))) float* pol_cfp;
        const pol_cfp w1 = (pol_cfp)(uintptr_t)a.pol.w[1];
        for (int j0 = 0; j0 < W1; j0 += 8) {
          const float4 ba = *(const float4*)(ob1 + j0), bb = *(const float4*)(ob1 + j0 + 4);
          float c[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
          const pol_cfp rb = w1 + j0 * W0;
          for (int k0 = 0; k0 < W0; k0 += 8) {
            float w[8][8], h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
              for (int v = 0; v < 8; ++v) w[u][v] = rb[u * W0 + k0 + v];
#pragma unroll
            for (int v = 0; v < 8; ++v) h[v] = hcol[(k0 + v) * NT];
            // (measured and dropped: a scheduling barrier here, so that all eight scalar loads are in flight before the first multiply-add --
            //  64 weight SGPRs live at once spill into VGPR lanes: 27.0 against 24.5 us per step for 3-64-64-1)
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
              for (int u = 0; u < 8; ++u) asm("v_fmac_f32 %0, %1, %2" : "+v"(c[u]) : "s"(w[u][v]), "v"(h[v]));      // c = fma(w, h, c): the definition's fmaf
          }
          const float4 wa = *(const float4*)(ow2 + j0), wb = *(const float4*)(ow2 + j0 + 4);
          y = __fmaf_rn(wa.x, pol_act<ACT>(c[0]), y); y = __fmaf_rn(wa.y, pol_act<ACT>(c[1]), y); y = __fmaf_rn(wa.z, pol_act<ACT>(c[2]), y); y = __fmaf_rn(wa.w, pol_act<ACT>(c[3]), y);
          y = __fmaf_rn(wb.x, pol_act<ACT>(c[4]), y); y = __fmaf_rn(wb.y, pol_act<ACT>(c[5]), y); y = __fmaf_rn(wb.z, pol_act<ACT>(c[6]), y); y = __fmaf_rn(wb.w, pol_act<ACT>(c[7]), y);
        }
      } else if (SW) {
        typedef const __attribute__((address_space(4))) float* pol_cfp;
        const pol_cfp w1 = (pol_cfp)(uintptr_t)a.pol.w[1];
        for (int j0 = 0; j0 < W1; j0 += 4) {                           // four units: four independent chains over k, ascending k each
          const float4 bj = *(const float4*)(ob1 + j0);
          float c0 = bj.x, c1 = bj.y, c2 = bj.z, c3 = bj.w;
          const pol_cfp r0 = w1 + j0 * W0, r1 = r0 + W0, r2 = r1 + W0, r3 = r2 + W0;
          for (int k0 = 0; k0 < W0; k0 += 8) {
            float h[8], wa[8], wb[8], wc[8], wd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { wa[u] = r0[k0 + u]; wb[u] = r1[k0 + u]; wc[u] = r2[k0 + u]; wd[u] = r3[k0 + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = hcol[(k0 + u) * NT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              c0 = __fmaf_rn(wa[u], h[u], c0); c1 = __fmaf_rn(wb[u], h[u], c1); c2 = __fmaf_rn(wc[u], h[u], c2); c3 = __fmaf_rn(wd[u], h[u], c3);
            }
          }
          const float4 wo = *(const float4*)(ow2 + j0);
          y = __fmaf_rn(wo.x, pol_act<ACT>(c0), y); y = __fmaf_rn(wo.y, pol_act<ACT>(c1), y); y = __fmaf_rn(wo.z, pol_act<ACT>(c2), y); y = __fmaf_rn(wo.w, pol_act<ACT>(c3), y);
        }
      } else
      for (int j0 = 0; j0 < W1p; j0 += 8) {                            // eight units of the second layer: eight independent chains over k
        float c[8];
        { const float4 ba = *(const float4*)(ob1 + j0), bb = *(const float4*)(ob1 + j0 + 4);
          c[0] = ba.x; c[1] = ba.y; c[2] = ba.z; c[3] = ba.w; c[4] = bb.x; c[5] = bb.y; c[6] = bb.z; c[7] = bb.w; }
        const float* const rows = o1 + j0 * W0p;
        for (int k0 = 0; k0 < W0p; k0 += 4) {
          const float h0 = hcol[k0 * NT], h1 = hcol[(k0 + 1) * NT], h2 = hcol[(k0 + 2) * NT], h3 = hcol[(k0 + 3) * NT];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float4 w = *(const float4*)(rows + u * W0p + k0);
            c[u] = __fmaf_rn(w.w, h3, __fmaf_rn(w.z, h2, __fmaf_rn(w.y, h1, __fmaf_rn(w.x, h0, c[u]))));      // ascending k
          }
        }
        const float4 wa = *(const float4*)(ow2 + j0), wb = *(const float4*)(ow2 + j0 + 4);
        y = __fmaf_rn(wa.x, pol_act<ACT>(c[0]), y); y = __fmaf_rn(wa.y, pol_act<ACT>(c[1]), y); y = __fmaf_rn(wa.z, pol_act<ACT>(c[2]), y); y = __fmaf_rn(wa.w, pol_act<ACT>(c[3]), y);
        y = __fmaf_rn(wb.x, pol_act<ACT>(c[4]), y); y = __fmaf_rn(wb.y, pol_act<ACT>(c[5]), y); y = __fmaf_rn(wb.z, pol_act<ACT>(c[6]), y); y = __fmaf_rn(wb.w, pol_act<ACT>(c[7]), y);
      }
    }
    const float av = __fmaf_rn(out_scale, y, out_bias);
    const float action = (av < out_lo ? out_lo : (av > out_hi ? out_hi : av)) + 0.0f;      // (+ 0.0f: an exact zero leaves as +0)

    // ---- PhantomEnv.step for the pair (env.py:239-303 with the supply chain's closed form) -------------------------------------------
    int D = 0;                                                         // the shop's customers' order sizes summed, supply_chain.py:61-67
    if (EXO) {
      const uint8_t* row = a.io.exo + ((int64_t)t * a.B + b) * a.n_exo;
      for (int k = 0; k < K; ++k) D += (int)row[a.shop_cust_exo[c0 + k]];
    } else {
      rng_quad_block(quad, a.seed, genv, tick, s);
      if (small_k) {                                                   // (uniform) every shop of the wave has at most six customers: one word, its first K base-5 digits
        uint32_t y, jr;
        if (__builtin_expect(!rng_split(rng_pick(quad.w, tick), y, jr), 0)) y = rng_group_y(a.seed, genv, tick, s, 0, 1);      // probability 3.3e-6
        y -= __umul24((uint32_t)((float)y * inv_pK), pK);              // y mod 5^K (exact through f32: tests/test_host_logic.py)
        D = rng_digit_sum6(y);
      } else D = rng_orders_from_block(quad.w, a.seed, genv, tick, s, K, nullptr, nullptr);
    }
    const int req = dev_round_half_even(action), room = PHX_SHOP_MAX_STOCK - stock;      // decode_action :136-142 (the stock BEFORE the step)
    const int deliv = req < room ? req : room;
    const int sell = stock < D ? stock : D;                            // handle_order_request, order after order: sells what is left (:105-122)
    sales = sell; missed = D - sell;
    stock = stock - sell + deliv;                                      // handle_stock_response (:98-103): deliv <= 100 - stock
    delivered = deliv;
    const int t_ep = step + 1;
    const bool trunc = t_ep == a.num_steps;                            // truncations["__all__"], env.py:312-318
    float ob[3];
    encode(stock, sales, missed, ob);
    const float rw = (float)shop_reward(sales, stock);                 // compute_reward :147, rounded once to f32
    if (on) {                                                          // the trajectory row, rollout.py:361-389 (running pointers: one 64-bit add per plane and step)
      p_obs[0] = ob[0]; p_obs[1] = ob[1]; p_obs[2] = ob[2];
      *p_act = action;
      *p_rew = rw;
      if (has_ter) { *p_ter = 0; p_ter += total; }
      *p_tru = trunc ? 1 : 0;
      p_obs += total * 3; p_act += total; p_rew += total; p_tru += total;
    }
    ++tick;
    if (trunc) {                                                       // the caller's env.reset(): ShopAgent.reset zeroes the stock (:149-150); sales stay (App. B)
      stock = 0; step = 0;
      encode(0, sales, missed, x);
    } else { step = t_ep; x[0] = ob[0]; x[1] = ob[1]; x[2] = ob[2]; }
  }
  if (on) {
    a.stock[pair] = stock; a.sales[pair] = sales; a.missed[pair] = missed; a.delivered[pair] = delivered;
    if (a.io.last_obs) { float* lo = a.io.last_obs + pair * 3; lo[0] = x[0]; lo[1] = x[1]; lo[2] = x[2]; }
    if (s == 0) { a.env_step[b] = step; a.env_tick[b] = (int32_t)tick; }
  }
}

// host: serves the call?  (plain supply chain on the fused schedule, ShopAgent observations, whole envs in a 256-lane workgroup)
const char* phx_sc_policy_unsupported(const DevSpec& sp, const phx_rollout_io& io) {
  const phx_policy_mlp& p = *io.policy;
  if (sp.env_type != PHX_ENV_PLAIN || sp.any_typed || sp.D != 3 || sp.S < 1 || sp.S > 128 || sp.S != sp.kind_count[PHX_KIND_SHOP]) return "the policy kernel serves plain supply-chain envs (ShopAgent observations, at most 128 shops)";
  if (p.n_hidden < 1 || p.n_hidden > 2) return "phx_policy_mlp: 1 or 2 hidden layers";
  for (int l = 0; l < p.n_hidden; ++l) if (p.width[l] < 1 || p.width[l] > PHX_POLICY_MAX_WIDTH) return "phx_policy_mlp: hidden widths 1 .. 64";
  if (p.activation != PHX_ACT_RELU && p.activation != PHX_ACT_HARD_TANH) return "phx_policy_mlp: unknown activation";
  for (int l = 0; l <= p.n_hidden; ++l) if (!p.w[l] || !p.b[l] || ((uintptr_t)p.w[l] & 3u) || ((uintptr_t)p.b[l] & 3u)) return "phx_policy_mlp: a weight / bias pointer is NULL or misaligned";
  if (!(p.out_lo >= 0.0f) || !(p.out_hi >= p.out_lo)) return "phx_policy_mlp: 0 <= out_lo <= out_hi (ShopAgent's action space)";
  return nullptr;
}

hipError_t phx_launch_sc_rollout_policy(const DevSpec& sp, const phx_rollout_io& io, hipStream_t st) {
  PolArgs a; memset(&a, 0, sizeof a);
  a.B = sp.B; a.S = sp.S; a.T = io.T; a.num_steps = sp.num_steps; a.n_exo = sp.n_exo;
  a.seed = sp.seed; a.env_offset = sp.env_offset;
  a.stock = (int32_t*)sp.f[F_SHOP_STOCK]; a.sales = (int32_t*)sp.f[F_SHOP_SALES]; a.missed = (int32_t*)sp.f[F_SHOP_MISSED];
  a.delivered = (int32_t*)sp.f[F_SHOP_DELIVERED]; a.env_step = (int32_t*)sp.f[F_ENV_STEP]; a.env_tick = (int32_t*)sp.f[F_ENV_TICK];
  a.shop_norm = sp.shop_norm; a.shop_cust_ptr = sp.shop_cust_ptr; a.shop_cust_exo = sp.shop_cust_exo;
  a.io = io; a.pol = *io.policy;
  const bool two = a.pol.n_hidden == 2;
  const int NT = two ? 128 : 256;
  a.epb = NT / sp.S;
  const dim3 grid((unsigned)((sp.B + a.epb - 1) / a.epb));
  const int W0p = (a.pol.width[0] + 7) & ~7, W1p = two ? (a.pol.width[1] + 7) & ~7 : 0;
  const int n_img = pol_img_floats(a.pol.n_hidden, W0p, W1p);
  const size_t lds = (size_t)n_img * 4 + (two ? (size_t)W0p * NT * sizeof(float) : 0);      // <= 18.7 + 32 KB
  phx_note_kernel("phx_sc_rollout_policy_kernel");
  const bool sw = two && (a.pol.width[0] & 7) == 0 && (a.pol.width[1] & 3) == 0;
#define POL_LAUNCH(ACT_, TWO_, NT_) do { \
    if (TWO_ && sw) { if (io.exo) hipLaunchKernelGGL((phx_sc_rollout_policy_kernel<ACT_, TWO_, NT_, true, TWO_>), grid, dim3(NT_), lds, st, a); \
                      else hipLaunchKernelGGL((phx_sc_rollout_policy_kernel<ACT_, TWO_, NT_, false, TWO_>), grid, dim3(NT_), lds, st, a); } \
    else if (io.exo) hipLaunchKernelGGL((phx_sc_rollout_policy_kernel<ACT_, TWO_, NT_, true, false>), grid, dim3(NT_), lds, st, a); \
    else hipLaunchKernelGGL((phx_sc_rollout_policy_kernel<ACT_, TWO_, NT_, false, false>), grid, dim3(NT_), lds, st, a); } while (0)
  if (a.pol.activation == PHX_ACT_HARD_TANH) { if (two) POL_LAUNCH(PHX_ACT_HARD_TANH, true, 128); else POL_LAUNCH(PHX_ACT_HARD_TANH, false, 256); }
  else { if (two) POL_LAUNCH(PHX_ACT_RELU, true, 128); else POL_LAUNCH(PHX_ACT_RELU, false, 256); }
#undef POL_LAUNCH
  return hipGetLastError();
}
