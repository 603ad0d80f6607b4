#!/bin/bash
# The reference's own programs, compiled unchanged against this repository's headers (oracle/_ref/, built in the build container), on the GPU box:
# benchmark/main.cpp with the 7.2 M-triangle stand-in written where it looks for assets/xyzrgb_dragon.obj, and demo/main.cpp with a wall-time split.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $out
work=/tmp/ref_programs; rm -rf $work; mkdir -p $work/assets
export PYTHONUNBUFFERED=1
t0=$(date +%s.%N)
python3 tools/write_standin_obj.py $work/assets/xyzrgb_dragon.obj 1900 > $out/standin_obj.log 2>&1 || { echo "writing the stand-in failed"; exit 1; }
t1=$(date +%s.%N)
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/cpupathtrace_amd:$LD_LIBRARY_PATH
(cd $work && timeout -k 10 600 $GRAFT_REPO_ROOT/oracle/_ref/ref_benchmark --benchmark_min_time=2 > $out/reference_benchmark_program.txt 2>&1) || { echo "ref_benchmark failed"; tail -5 $out/reference_benchmark_program.txt; }
t2=$(date +%s.%N)
(cd $work && PATHTRACE_SEED=1234 PT_DEBUG=1 timeout -k 10 600 $GRAFT_REPO_ROOT/oracle/_ref/ref_demo $work/out/demo.png > $out/reference_demo_stdout.txt 2> $out/reference_demo_stderr.txt) || { echo "ref_demo failed"; tail -5 $out/reference_demo_stderr.txt; }
t3=$(date +%s.%N)
python3 - $t0 $t1 $t2 $t3 $out $work <<'PY'
import os, re, sys
t0, t1, t2, t3 = map(float, sys.argv[1:5])
out, work = sys.argv[5], sys.argv[6]
err = open(out + "/reference_demo_stderr.txt").read()
with open(out + "/reference_demo_timing.txt", "w") as f:
    f.write("demo/main.cpp (unchanged) with the 7,216,200-triangle stand-in mesh, 256 x 256, 16..64 spp, PATHTRACE_SEED=1234: %.2f s wall in all\n" % (t3 - t2))
    f.write("(writing the stand-in OBJ took %.1f s, the benchmark program %.1f s)\n" % (t1 - t0, t2 - t1))
    for line in err.splitlines():
        if line.startswith("[pt]") and ("scene build" in line or "path kernel:" in line or "wave time" in line):
            f.write(line + "\n")
    png = work + "/out/demo.png"
    f.write("output: %s, %d bytes\n" % (os.path.basename(png), os.path.getsize(png) if os.path.exists(png) else -1))
print(open(out + "/reference_demo_timing.txt").read())
PY
cat $out/reference_benchmark_program.txt | tail -5
