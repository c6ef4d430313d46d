#!/bin/bash
# Collect the round's judged profile artefacts on the GPU box (run through gpurun); outputs under
# gpurun_out/round_profiles/, to be copied into profiles/ by the caller.  usage: collect_profiles.sh TAG
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_profiles; rm -rf $OUT; mkdir -p $OUT
# 1. kernel-trace stats of the bench command itself (the judged average duration of the dominant kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --extras --steps 20 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
keep = [r for r in rows if "chain_f32" in r["Name"] or "q15" in r["Name"] or "q7" in r["Name"]]
with open(f"{out}/{tag}_kernel_stats.csv", "w") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"])):
        w.writerow([r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
PY
# 2. un-profiled bench line (with the CPU baseline)
python3 bench.py --extras > $OUT/${TAG}_bench.json 2> $OUT/bench.err
# 3. PMC passes: fused float kernel (IIR, bypass) and the Q15 kernels
bash tools/pmc_profile.sh 0xA1 round_profiles/pmc_iir > $OUT/${TAG}_pmc_chain_f32_iir.txt 2>&1
bash tools/pmc_profile.sh 0xB1 round_profiles/pmc_byp > $OUT/${TAG}_pmc_chain_f32_noiir.txt 2>&1
bash tools/pmc_profile.sh q15 round_profiles/pmc_q15 > $OUT/${TAG}_pmc_q15.txt 2>&1
# 4. phase stamps and workgroup timeline (diagnostic build)
python3 tools/phase_stamps.py 4096 0xA1 > $OUT/${TAG}_phase_stamps_iir.txt 2>&1
python3 tools/phase_stamps.py 256 0xA1 > $OUT/${TAG}_phase_stamps_iir_lone.txt 2>&1
python3 tools/phase_stamps.py 4096 0xB1 > $OUT/${TAG}_phase_stamps_noiir.txt 2>&1
python3 tools/wg_timeline.py 4096 0xA1 > $OUT/${TAG}_wg_timeline.txt 2>&1
# 5. memory skeleton and ingest
./tools/ubench/frame_stream 4096 6 60 > $OUT/${TAG}_memory_skeleton_raw.txt 2>&1
python3 tools/ingest_bench.py 1024 32 0xB1 --events > $OUT/${TAG}_ingest_raw.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_iir $OUT/pmc_byp $OUT/pmc_q15
ls -la $OUT
