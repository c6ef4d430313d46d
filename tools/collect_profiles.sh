#!/bin/bash
# Collect the round's judged profile artefacts on the GPU box (run through gpurun); outputs under
# gpurun_out/round_profiles/, to be copied into profiles/ by the caller.
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_profiles; rm -rf $OUT; mkdir -p $OUT
# 1. kernel-trace stats of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --extras --steps 20 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
# 2. un-profiled bench line (with the CPU baseline)
python3 bench.py --extras > $OUT/${TAG}_bench.json 2> $OUT/bench.err
# 3. PMC passes for the dominant kernel and the no-IIR kernel
bash tools/pmc_profile.sh 0xA1 round_profiles/pmc_iir > $OUT/${TAG}_pmc_chain_f32_iir.txt 2>&1
bash tools/pmc_profile.sh 0xB1 round_profiles/pmc_byp > $OUT/${TAG}_pmc_chain_f32_noiir.txt 2>&1
# 4. phase stamps (diagnostic build)
python3 tools/phase_stamps.py 4096 0xA1 > $OUT/${TAG}_phase_stamps_iir.txt 2>&1
python3 tools/phase_stamps.py 4096 0xB1 > $OUT/${TAG}_phase_stamps_noiir.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_iir $OUT/pmc_byp
ls -la $OUT
