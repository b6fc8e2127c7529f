#!/bin/bash
# Re-creates the round-2 "hipFree never returns" report under the watchdog (profiles/r03_hang_diagnosis.txt):
# the engine-owned stream for NULL-stream callers is switched back on, the pool sets device-wide blocking sync, and the
# harness tests run as a whole.  When a runtime wait overstays, every thread prints its host backtrace, every engine stream
# answers a hipStreamQuery and the process exits with 86 (RGBD_DIAG_EXIT) instead of sitting on the box.
#   RGBD_BLOCKING_SYNC=0 bash tools/hang_repro.sh      -> the same sequence completes (spinning waits)
set -o pipefail
mkdir -p gpurun_out
RGBD_DIAG_EXIT=1 RGBD_NULL_OWN_STREAM=1 RGBD_DEBUG_DESTROY=1 timeout -k 10 300 \
  python -m pytest tests/test_gpu_harness.py -x -q -s -p no:cacheprovider 2>&1 | tee gpurun_out/hang_repro.log | grep -a "watchdog\]\|passed\|failed"
echo "exit code ${PIPESTATUS[0]} (86 = a watched wait overstayed; backtraces in gpurun_out/hang_repro.log)"
