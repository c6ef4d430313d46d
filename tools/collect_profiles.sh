#!/bin/bash
# Collect the round's judged profile artefacts on the GPU box (run through gpurun); outputs under
# gpurun_out/round_profiles/, to be copied into profiles/ by the caller.  usage: collect_profiles.sh TAG
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_profiles; rm -rf $OUT; mkdir -p $OUT
# 1. kernel-trace stats of the bench command itself, in two SEPARATE processes so that no kernel's average mixes modes:
#    `--overlap 1 --extras` launches every kernel of the product strictly stream-ordered (float chain, bypass, both
#    integer cascades, the wide cascade, the integer FFT) -> ${TAG}_kernel_stats.csv, the judged per-kernel durations;
#    the default line (two launches in flight) with its extras -> ${TAG}_kernel_stats_overlap2.csv (overlapped kernels
#    share the chip and last longer each).  --no-power: no side thread, no child process under the profiler.
stats() {   # stats TRACE_DIR OUT_CSV
python3 - "$1" "$2" <<'PY'
import csv, glob, sys
src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(src + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r["Name"] for k in ("chain_f32", "q15", "q7", "w14"))]
with open(dst, "w") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"])):
        nm = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        w.writerow([nm.split("(")[0] + (" [int16 samples in]" if "(short const*" in nm and "_f32_kernel" in nm else ""), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
PY
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 bench.py --overlap 1 --no-cpu-baseline --no-power --extras --steps 20 > $OUT/${TAG}_bench_ordered_under_rocprof.json 2> $OUT/bench_ordered_under_rocprof.err
stats $OUT/trace1 $OUT/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py --no-cpu-baseline --no-power --extras --steps 20 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
stats $OUT/trace2 $OUT/${TAG}_kernel_stats_overlap2.csv
# 2. un-profiled bench lines (with the CPU baseline): headline mode with the extras, ordered mode
python3 bench.py --extras > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python3 bench.py --overlap 1 --no-cpu-baseline > $OUT/${TAG}_bench_ordered.json 2> $OUT/bench_ordered.err
# 3. PMC passes: fused float kernel (IIR, bypass) and the Q15 kernels
bash tools/pmc_profile.sh 0xA1 round_profiles/pmc_iir > $OUT/${TAG}_pmc_chain_f32_iir.txt 2>&1
bash tools/pmc_profile.sh 0xB1 round_profiles/pmc_byp > $OUT/${TAG}_pmc_chain_f32_noiir.txt 2>&1
bash tools/pmc_profile.sh q15 round_profiles/pmc_q15 > $OUT/${TAG}_pmc_q15.txt 2>&1
SA_PMC_SHORT=1 bash tools/pmc_profile.sh q15wide round_profiles/pmc_q15wide > $OUT/${TAG}_pmc_q15_wide.txt 2>&1
SA_PMC_SHORT=1 bash tools/pmc_profile.sh 0xA1 round_profiles/pmc_spec spec_half > $OUT/${TAG}_pmc_chain_f32_spec_half.txt 2>&1
python3 - "$OUT" "$TAG" <<'PY'
# HBM traffic per launch of the headline kernel for bench.py's roofline.traffic (FETCH_SIZE doubled: gfx950 counts the
# 128-byte requests of wide coalesced streams at 64 bytes, MI355X_MICROARCH.md, HBM section; WRITE_SIZE exact)
import json, re, sys
out, tag = sys.argv[1], sys.argv[2]
txt = open(f"{out}/{tag}_pmc_chain_f32_iir.txt").read()
f = float(re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean\s+([\d.]+)", txt).group(1))
w = float(re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean\s+([\d.]+)", txt).group(1))
k = re.search(r"== (chain_f32_kernel<[^>]*>)", txt).group(1)
# secondary ceiling: how busy the vector pipes were.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over
# every SIMD (256 CUs x 4), GRBM_GUI_ACTIVE cycles summed over the 8 XCDs (MI355X_MICROARCH.md, counter units)
va = float(re.search(r"SQ_ACTIVE_INST_VALU\s+n=\s*\d+\s+mean\s+([\d.]+)", txt).group(1))
ga = float(re.search(r"GRBM_GUI_ACTIVE\s+n=\s*\d+\s+mean\s+([\d.]+)", txt).group(1))
valu_busy = (4.0 * va / 1024.0) / (ga / 8.0)
json.dump({"kernel": k, "batch": 4096,
           "source": f"profiles/{tag}_pmc_chain_f32_iir.txt (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE, 4 launches each)",
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-byte streaming stores",
           "traffic_bytes_per_launch": int(round((2 * f + w) * 1024)),
           "algorithmic_bytes_per_launch": 4096 * 131072,
           "SQ_ACTIVE_INST_VALU": va, "GRBM_GUI_ACTIVE": ga, "valu_busy_frac": round(valu_busy, 4),
           "valu_busy_note": "4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs: share of the launch a SIMD's vector pipe is executing"},
          open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
PY
# 4. phase stamps and workgroup timeline (diagnostic build)
python3 tools/phase_stamps.py 4096 0xA1 > $OUT/${TAG}_phase_stamps_iir.txt 2>&1
python3 tools/phase_stamps.py 256 0xA1 > $OUT/${TAG}_phase_stamps_iir_lone.txt 2>&1
python3 tools/phase_stamps.py 4096 0xB1 > $OUT/${TAG}_phase_stamps_noiir.txt 2>&1
python3 tools/phase_stamps.py 256 0xB1 > $OUT/${TAG}_phase_stamps_noiir_lone.txt 2>&1
python3 tools/wg_timeline.py 4096 0xA1 > $OUT/${TAG}_wg_timeline.txt 2>&1
# 5. memory skeleton and ingest
./tools/ubench/frame_stream 4096 6 60 > $OUT/${TAG}_memory_skeleton_raw.txt 2>&1
python3 tools/ingest_bench.py 1024 32 0xB1 --events > $OUT/${TAG}_ingest_raw.txt 2>&1
# (package power and shader clock: bench.py samples them in process from sysfs next to the headline -- roofline.power in
#  ${TAG}_bench.json; the round-3 study is profiles/r3_power_clock.txt, tools/power_clock.py)
rm -rf $OUT/trace1 $OUT/trace2 $OUT/pmc_iir $OUT/pmc_byp $OUT/pmc_q15 $OUT/pmc_q15wide $OUT/pmc_spec
ls -la $OUT
