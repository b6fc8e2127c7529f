#!/usr/bin/env python3
"""Host CPU seconds a command burns per wall second (how many cores one bench rank needs): python tools/cpu_usage.py <cmd...>"""
import os
import subprocess
import sys
import time

t0 = time.time()
c0 = os.times()
rc = subprocess.call(sys.argv[1:])
c1 = os.times()
wall = time.time() - t0
cpu = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
print(f"wall {wall:.1f} s, cpu {cpu:.1f} s (user {c1.children_user - c0.children_user:.1f}, sys {c1.children_system - c0.children_system:.1f}) "
      f"-> {cpu / wall:.2f} cores busy on average", file=sys.stderr)
sys.exit(rc)
