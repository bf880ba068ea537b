#!/bin/bash
# SQ counters of the feature stage's heavy kernels (separate --pmc passes, program directly after `--`):
#   bash tools/freg_counters.sh -> gpurun_out/freg_counters.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/freg_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_WAVES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/p$i --output-format csv -- python3 tools/bench_extract.py 2048 3 > /dev/null 2> $OUT/p$i.err
done
python3 - "$OUT" <<'PY' | tee gpurun_out/freg_counters.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in sorted(glob.glob(out + "/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("smooth_slide_kernel<24", "smooth_slide_kernel<12", "km_shortlist", "kp_sort_kernel", "rs_refine", "nmi_reduce_kernel", "knn2_kernel", "daisy_sample_kernel") if k in n), None)
        if key:
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    d = acc[k]
    print(k)
    for c in sorted(d):
        print(f"   {c:34s} {d[c]:.5g}")
    g = lambda c: d.get(c, 0.0)
    simd_cycles = 1024 * g("GRBM_GUI_ACTIVE") / 8
    if simd_cycles:
        print(f"   VALU busy share of SIMD cycles     {4 * g('SQ_ACTIVE_INST_VALU') / simd_cycles:.3f}")
        print(f"   waves resident per SIMD (avg)      {g('SQ_WAVE_CYCLES') * 4 / simd_cycles:.2f}")
        print(f"   VALU instr per wave                {g('SQ_INSTS_VALU') / max(g('SQ_WAVES'), 1):.0f}")
        if g('SQ_INSTS_VMEM'):
            print(f"   VMEM latency (cycles, avg)         {g('SQ_INST_LEVEL_VMEM') / g('SQ_INSTS_VMEM'):.0f}")
        if g('TCC_REQ_sum'):
            print(f"   L2 hit rate                        {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.3f}")
PY
