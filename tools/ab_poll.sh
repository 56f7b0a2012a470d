#!/bin/bash
# GPU box: completion by flag (default) against the runtime's stream wait (SBE_POLL_DONE=0), alternating on one box:
# GPU suite once, then the call-log replay, the 64-chain batched step and the single-step timings.  (profiles/r3/ab_poll_done.log)
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r3x.log 2>&1; tail -2 gpurun_out/gputest_r3x.log
for i in 1 2 3; do
  for p in 0 1; do
    echo "poll=$p $(SBE_POLL_DONE=$p python tools/replay_bench.py headline 2>/dev/null | cut -c1-95)"
  done
done
for p in 0 1 0 1; do echo "poll=$p $(SBE_POLL_DONE=$p python tools/replay_bench.py cfg1 south_america 2>/dev/null | cut -c1-95 | tr '\n' ' ')"; done
for p in 0 1 0 1; do echo "poll=$p $(SBE_POLL_DONE=$p python tools/prof_step_batch.py 2>&1 | tail -2 | cut -c1-110 | tr '\n' ' ')"; done
for p in 0 1 0 1; do echo "poll=$p $(SBE_POLL_DONE=$p python tools/prof_step.py 2>&1 | head -10 | tr '\n' ';')"; done
