"""What bench.py is made of (bench.py itself is the driver's entry point and only orchestrates):
  args.py      the command line
  launch.py    `python bench.py --gpus N` without a launcher: N fresh child ranks, started before any GPU call
  rollout.py   the timed closed-loop rollout of one workload (+ new scenes through the same engine)
  training.py  the timed training_step (16 scenes per GPU; gradients all-reduced over RCCL when world > 1)
  events.py    per-kernel-class HIP-event passes (the `roofline` objects) + the committed PMC passes' traffic
  cpu.py       the oracle on the host cores (`cpu_baseline`)
  report.py    the ONE judged JSON line (compact, < 8 KB) and the detail side file
"""
