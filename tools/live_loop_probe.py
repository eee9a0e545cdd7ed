"""Runs host/bench/live_loop (the C++ drivers around the C ABI with a host oracle) on BASELINE's synthetic cuts.
usage: python tools/live_loop_probe.py [n] [warm] [steps] [workload: parallel|deep]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ellalgo_rs_amd as pkg
from ellalgo_rs_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 48
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
gen = synth.deep_cuts if (len(sys.argv) > 4 and sys.argv[4] == "deep") else synth.parallel_cuts
exe = pkg.build.build_host_tools()
kinds, grads, b0, b1 = gen(n, warm + steps + 1)
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    path = os.path.join(d, "cuts.bin")
    synth.write_cuts_bin(path, kinds, grads, b0, b1)
    del grads
    out = subprocess.run([exe, path, str(warm), str(steps)], capture_output=True, text=True)
    print(out.stdout.strip())
    if out.returncode:
        print("rc", out.returncode, out.stderr, file=sys.stderr)
        sys.exit(out.returncode)
