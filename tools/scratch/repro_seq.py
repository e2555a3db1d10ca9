"""Replays the GPU rollout tests as plain function calls (no pytest) followed by the rule-buffer test."""
import os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
tb = load_package()
import test_hip_rollout as R
import test_hip_rules as U
gd = ROOT / "tests" / "golden"
seq = [lambda: R.test_reactive_replay_vs_oracle_and_reference(tb, gd, (8, 64, 8), 4, 90, "c1"),
       lambda: R.test_reactive_replay_vs_oracle_and_reference(tb, gd, (64, 1024, 128), 32, 14, "c2"),
       lambda: R.test_teacher_forced_replay(tb, (8, 64, 8), 4, 90),
       lambda: R.test_teacher_forced_replay(tb, (64, 1024, 128), 32, 24),
       lambda: R.test_free_rollout_90_steps_damped_policy(tb),
       lambda: R.test_joint_future_pred_shares_map_and_matches_single(tb),
       lambda: R.test_lights_one_step_ahead_equals_sequential_order(tb),
       lambda: R.test_hoisted_rollout_constants_are_bit_identical(tb),
       lambda: U.test_rollout_buffer_carries_rule_violations(tb)]
for i, f in enumerate(seq):
    f()
    print("step", i, "ok", flush=True)
print("sequence survived")
