#!/bin/bash
# Per-kernel SQ / cache counters (separate --pmc passes, program directly after `--`):
#   bash tools/sq_counters.sh [workload] [tag]      -> gpurun_out/sq_<workload>_<tag>/summary.txt
# Companion stream off (every launch alone on the chip).  The two instantiations of fb_blur_h_solve (LAST = false / true)
# are listed separately.  MICROALIGNER_HIP_LIB selects another build.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=${1:-cfg3}; TAG=${2:-tree}
# a fresh directory per invocation: results of earlier invocations (gpurun merges them into the caller's gpurun_out/) must never
# be summed into this one's (round 5's "library C" section was three runs added up: counts x 3, a 6 GHz clock)
OUT=gpurun_out/sq_${WL}_${TAG}_$(date +%s)_$$
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-companion"
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_IFETCH" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/p$i --output-format csv -- $CMD > /dev/null 2> $OUT/p$i.err
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(lambda: [0, 0.0])
grbm_ms = collections.defaultdict(float)
seen = set()
for f in sorted(glob.glob(out + "/p*/*/*_counter_collection.csv")):
    pas = f.split("/")[-3]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("fb_blur_h_solve", "fb_blur_v_stream", "fb_blur_v", "fb_polyexp_m0", "dog_fused", "scale_to_u8",
                                "joint_hist16_kernel", "nmi_reduce_kernel", "warp_tiled_kernel", "merge_flows_kernel") if k in n), None)
        if not key:
            continue
        if key == "fb_blur_h_solve":
            m = re.search(r"fb_blur_h_solve<([^>]*)>", n)
            args = [a.strip() for a in m.group(1).split(",")] if m else []
            key += " LAST=" + (args[4] if len(args) > 4 else "?")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        tag = (pas, r.get("Dispatch_Id"))
        if pas.endswith("p1") and tag not in seen and "End_Timestamp" in r:
            seen.add(tag)
            dur[key][0] += 1; dur[key][1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
        # the clock comes from ONE pass: the GRBM pass's own cycle counts over its own launch durations
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and tag not in seen and "End_Timestamp" in r:
            seen.add(tag)
            grbm_ms[key] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
for k in sorted(acc):
    d = acc[k]
    n, ms = dur[k]
    print(f"{k}   (launches in one pass: {n}, summed duration under the counter pass {ms:.2f} ms)")
    for c in sorted(d):
        print(f"   {c:34s} {d[c]:.5g}")
    g = lambda c: d.get(c, 0.0)
    if g("SQ_BUSY_CU_CYCLES"):
        simd_cycles = 1024 * g("GRBM_GUI_ACTIVE") / 8      # 256 CUs x 4 SIMDs x the kernel's cycles (GRBM is summed over 8 XCDs)
        if simd_cycles:
            print(f"   -- VALU-busy: SIMD cycles executing a VALU instruction   {g('SQ_ACTIVE_INST_VALU') * 4 / simd_cycles:.3f}   (ACTIVE_INST_* count quad-cycles summed over waves)")
            print(f"   -- waves resident per SIMD (of 8 slots)     {g('SQ_WAVE_CYCLES') * 4 / simd_cycles:.2f}")
            gm = grbm_ms.get(k, 0.0)
            print(f"   -- clock                                    {g('GRBM_GUI_ACTIVE') / 8 / (gm * 1e6) if gm else 0:.3f} GHz (cycles and durations of the GRBM pass)")
        print(f"   -- wave cycles waiting on an instruction    {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f} of wave cycles; waiting on anything {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):.3f}")
        print(f"   -- VMEM instructions in flight per wave     {g('SQ_INST_LEVEL_VMEM') / max(g('SQ_WAVE_CYCLES'), 1):.3f}")
        print(f"   -- LDS bank-conflict share of LDS cycles    {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.4f}")
        print(f"   -- TA address FIFO full / busy CU cycles    {g('SQ_VMEM_TA_ADDR_FIFO_FULL') * 4 / g('SQ_BUSY_CU_CYCLES'):.3f}")
    if g("TCC_HIT_sum") + g("TCC_MISS_sum"):
        print(f"   -- L2 hit rate                              {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
PY
cp $OUT/summary.txt gpurun_out/sq_${WL}_${TAG}.txt
