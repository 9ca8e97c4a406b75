"""Counts the instructions of every kernel of extend_budget.s by issue class (profiles/r03_valu_issue_microbench.txt: full-rate VALU 2.2 cycles
per wave64 instruction, half rate 4.2, quarter rate 8.2) and prints the per-block budget of wf_extend. Usage: extend_budget.py extend_budget.s"""
import re, sys
FULL = ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_mov_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_not_b32", "v_fma_legacy", "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32", "v_accvgpr")
QUARTER = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag")
def classify(op):
    if op.startswith(QUARTER): return "quarter"
    if op.startswith(FULL): return "full"
    return "half"
kern = None; rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(budget_\w+):", line)
    if m: kern = m.group(1); rows[kern] = {"full": 0, "half": 0, "quarter": 0, "salu": 0, "lds": 0, "vmem": 0, "half_ops": {}}; continue
    if kern is None: continue
    if re.match(r"^\s*s_endpgm", line): kern = None; continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if not m: continue
    op = m.group(1); r = rows[kern]
    if op.startswith("v_"):
        c = classify(op); r[c] += 1
        if c == "half": key = re.sub(r"_e(32|64)$", "", op); r["half_ops"][key] = r["half_ops"].get(key, 0) + 1
    elif op.startswith("ds_"): r["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): r["vmem"] += 1
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_load", "s_clause")): r["salu"] += 1
print("block (instructions as compiled; every kernel also holds the loads / stores of its inputs and results: 2-4 VMEM + the address arithmetic, ~10 VALU)")
print(f"{'':34s} {'full':>5s} {'half':>5s} {'quart':>5s} | {'VALU':>5s} {'pipe cycles: full / half+quarter':>34s} | {'SALU':>5s} {'LDS':>4s} {'VMEM':>5s}   half-rate instructions")
for k, r in rows.items():
    valu = r["full"] + r["half"] + r["quarter"]
    top = ", ".join(f"{n} x{c}" for n, c in sorted(r["half_ops"].items(), key=lambda kv: -kv[1])[:7])
    print(f"{k:34s} {r['full']:5d} {r['half']:5d} {r['quarter']:5d} | {valu:5d} {r['full'] * 2.2:14.0f} / {r['half'] * 4.2 + r['quarter'] * 8.2:<17.0f} | {r['salu']:5d} {r['lds']:4d} {r['vmem']:5d}   {top}")
