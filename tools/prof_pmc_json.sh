#!/bin/bash
# Counter-derived per-launch numbers for bench.py's roofline block -> gpurun_out/${TAG:-r06}_pmc.json (copy to profiles/).
#   HBM bytes     : reads  = 32 x TCC_EA0_RDREQ_32B + 64 x TCC_EA0_RDREQ_64B + 128 x TCC_EA0_RDREQ_128B (the L2's memory-side
#                   requests by size; calibrated on known byte counts in this code's access patterns --
#                   profiles/r03_traffic_calib.txt: every pattern, streams and record gathers alike, is served by 128-byte
#                   requests, which FETCH_SIZE tallies at 64 B: FETCH_SIZE x 2 is kept beside it as a cross-check);
#                   writes = 1024 x WRITE_SIZE (calibrated x1).  SEPARATE rocprofv3 --pmc passes, kernel dispatch only.
#                   Requests served by the 256 MB Infinity Cache are counted too (a cache-resident table still shows).
#   FP64 VALU     : SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 (wave-level instructions: x64 lanes, FMA = 2 flops)
#   VALU / LDS    : SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT against
#                   SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES
# for the matched default workload and for the exhaustive mode of the same scene.  The file records the hash of
# the device sources; bench.py quotes the numbers only while that hash matches.
# usage (on the GPU box, repo root): bash tools/prof_pmc_json.sh
repo=$(pwd)
export TMPDIR=/tmp
mkdir -p $repo/gpurun_out
groups=(
"FETCH_SIZE"
"WRITE_SIZE"
"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
"SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
"SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
)
for mode in matched exhaustive; do
  i=0
  : > $repo/gpurun_out/pmcj_$mode.csv
  for g in "${groups[@]}"; do
    i=$((i+1))
    cd /tmp && rm -rf /tmp/pmcj_${mode}_$i
    steps=3; [ $mode = exhaustive ] && steps=2
    timeout 150 rocprofv3 --pmc $g -d /tmp/pmcj_${mode}_$i -- python $repo/bench.py --steps $steps --warmup 1 --mode $mode --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pmcj_${mode}_$i.err || { echo "# group $i failed: $g"; tail -3 /tmp/pmcj_${mode}_$i.err; continue; }
    db=$(find /tmp/pmcj_${mode}_$i -name "*.db" | head -1)
    python $repo/tools/rocpd_pmc.py $db 2>/dev/null >> $repo/gpurun_out/pmcj_$mode.csv
  done
done
cd $repo && python - <<'PY'
import csv, json, re, sys
sys.path.insert(0, ".")
import bench
out = {"device_source_hash": bench.device_source_hash(),
       "workload": "bench.py default scene (100 views x 500 segs, nn 20): matched topk 10 / exhaustive",
       "how": "tools/prof_pmc_json.sh: one rocprofv3 --pmc pass per counter group, kernel dispatch only; averages per launch",
       "units": {"hbm_bytes": "bytes per launch = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B (L2 memory-side read requests by "
                              "size, calibration profiles/r03_traffic_calib.txt) + 1024 x WRITE_SIZE; hbm_bytes_fetchsize = 2 x 1024 x "
                              "FETCH_SIZE + 1024 x WRITE_SIZE (the guide's correction) as a cross-check",
                 "valu_flops_f64": "64 x (ADD + MUL + TRANS) + 128 x FMA wave-level FP64 instructions",
                 "valu_insts": "SQ_INSTS_VALU (wave-level; x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time) = share of the VALU issue slots)",
                 "lds_active_frac": "SQ_ACTIVE_INST_LDS / SQ_BUSY_CU_CYCLES"},
       "kernels": {}}
# (since round 5: the line-slot form's kernels are what runs on the bench scene; bench.py asks for the stage names)
alias = {"k_gen_ex_block": "k_gen_exhaustive", "k_gates_ln": "k_gates", "k_tri_rounds": "k_tri_rows", "k_place_rounds": "k_place"}
for mode in ("matched", "exhaustive"):
    vals = {}
    try:
        rows = list(csv.reader(open(f"gpurun_out/pmcj_{mode}.csv")))
    except FileNotFoundError:
        continue
    for row in rows:
        if len(row) != 5 or row[0] == "kernel":
            continue
        m = re.search(r"(k_[a-z_0-9]+)", row[0])
        if not m:
            continue
        # (instantiations of one template -- k_depth_order<big> / <small> -- run once each per step: their counters add up)
        d = vals.setdefault(m.group(1), {})
        d[row[1]] = d.get(row[1], 0.0) + float(row[3])
    kern = {}
    for name, v in vals.items():
        k = {"raw": v}
        if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
            k["hbm_bytes_fetchsize"] = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0) + 1024.0 * v.get("WRITE_SIZE", 0.0)
            k["hbm_bytes"] = k["hbm_bytes_fetchsize"]
        if "TCC_EA0_RDREQ_128B_sum" in v:
            k["hbm_read_bytes"] = (32.0 * v.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * v.get("TCC_EA0_RDREQ_64B_sum", 0.0)
                                   + 128.0 * v["TCC_EA0_RDREQ_128B_sum"])
            k["hbm_bytes"] = k["hbm_read_bytes"] + 1024.0 * v.get("WRITE_SIZE", 0.0)
        if "SQ_INSTS_VALU_FMA_F64" in v:
            k["valu_flops_f64"] = 64.0 * (v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0)
                                          + v.get("SQ_INSTS_VALU_TRANS_F64", 0)) + 128.0 * v["SQ_INSTS_VALU_FMA_F64"]
            k["valu_insts"] = v.get("SQ_INSTS_VALU")
        if v.get("SQ_BUSY_CU_CYCLES"):
            k["valu_busy_frac"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_BUSY_CU_CYCLES"]
            k["lds_active_frac"] = v.get("SQ_ACTIVE_INST_LDS", 0.0) / v["SQ_BUSY_CU_CYCLES"]
        kern[alias.get(name, name)] = k
    # the generation stage of the exhaustive mode (one-pass form) is k_gates_ex + k_tri_ex: summed per run
    if "k_gates_ex" in kern and "k_tri_ex" in kern:
        a, b = kern["k_gates_ex"], kern["k_tri_ex"]
        kern["k_gen_exhaustive"] = {f: a[f] + b[f] for f in ("hbm_bytes", "valu_flops_f64", "valu_insts") if f in a and f in b}
    out["kernels"][mode] = kern
import os
json.dump(out, open("gpurun_out/%s_pmc.json" % os.environ.get("TAG", "r06"), "w"), indent=1)
for mode, kern in out["kernels"].items():
    for n in ("k_score_q", "k_score3", "k_dense8", "k_gates", "k_tri_rows", "k_place", "k_gen_exhaustive"):
        if n in kern:
            print(mode, n, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in kern[n].items() if a != "raw"})
PY
