#!/usr/bin/env python
"""Golden of a STATE-DEPENDENT FSM stage handler, by running the reference (build container only; see gen_goldens.py).

The RESTOCK stage's handler does what fsm.py:294-302 documents: it calls ``self.resolve_network()`` -- so the shops' stock
requests of THIS step have been delivered -- and then branches on agent state: restock again while the shops together hold
fewer than 60 items, sell otherwise.  A handler evaluated before the step's acting phase (round 3's host callback) would see
the stock before the delivery and choose differently: the golden pins the reference's ordering (VERDICT r3 Missing #2).

    python tests/golden/gen_goldens_fsm_state.py      # rewrites tests/golden/sc_fsm_state_handler.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import gen_goldens as gg  # noqa: E402

THRESHOLD = 60


def stock_handler(env):
    env.resolve_network()
    total = sum(a.stock for aid, a in env.agents.items() if aid.startswith("SHOP"))
    return "RESTOCK" if total < THRESHOLD else "SELL"


def act(t, b, s):
    return ((t * 37 + b * 11 + s * 5) % 23) * 1.37 + 0.5 * (s == 1)      # small requests: several RESTOCK steps in a row


if __name__ == "__main__":
    gg.run_supply_chain("sc_fsm_state_handler", 3, [2, 4, 3], 30, 70, [61, 62, 63], act, norm_customers=4,
                        fsm=True, log_steps=2, handler=stock_handler)
