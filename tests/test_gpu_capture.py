"""The hipGraph capture contract (engine.CaptureFailed): a failed capture is reported and ends the process with exit code 3 -- it is never
`recovered' from in-process (round 2's recovery path crashed with SIGSEGV a few seconds later) -- and bench.py probes the capture in a
child process before the measuring process touches the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_failed_capture_ends_the_process_with_the_reason_and_code_3():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capture_failure_child.py")], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "hipGraph capture failed" in r.stderr and "trainer refuses further launches" in r.stderr, r.stderr[-2000:]


def test_bench_probes_the_capture_in_a_child_process_first():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "hipGraph replay" in line["config"]["workload"], line["config"]
    assert "capture probe ok" in r.stderr          # the child's verdict (its stderr is inherited)
