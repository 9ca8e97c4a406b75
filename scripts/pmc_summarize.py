"""Summarises the rocprofv3 --pmc passes of scripts/gpu_r03_pmc.sh: per kernel class VALU issue figures, LDS figures and HBM traffic per launch.
Writes <out>/pmc_valu_config<N>.json, pmc_traffic_config<N>.json and a text table."""
import collections
import csv
import glob
import json
import re
import sys

out, config = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(wf_shadow_rays|wf_store_primary|wf_[a-z]+|pt_megakernel)(<[^>]*>)?", row["Kernel_Name"])
        if not m:
            continue
        name = m.group(1)
        targs = [a.strip() for a in m.group(2).strip("<>").split(",")] if m.group(2) else []
        # wf_extend<LDS, DEPTH, W, ANYHIT, TL, PRIMARY>, wf_shade<MAXL, SIMPLE, PRIMARY>: the any-hit pass and the bounce-0 instantiations get their own rows
        if name == "wf_extend" and len(targs) > 3 and targs[3] in ("true", "1"):
            name = "wf_extend_anyhit"
        elif name == "wf_extend" and len(targs) > 5 and targs[5] in ("true", "1"):
            name = "wf_extend_primary"
        elif name == "wf_shade" and len(targs) > 2 and targs[2] in ("true", "1"):
            name = "wf_shade_primary"
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(wf_shadow_rays|wf_store_primary|wf_[a-z]+|pt_megakernel)", row["Kernel_Name"])
        if m:
            dur[m.group(1)].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6)
CUS, SIMDS = 256, 1024
valu, traffic, lines = {}, {}, []
for k, cs in sorted(agg.items()):
    g = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in g or "WRITE_SIZE" in g:
        # rocprofv3 reports KB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM)
        traffic[k] = {"fetch_size_kb_raw": g.get("FETCH_SIZE", 0.0), "write_size_kb": g.get("WRITE_SIZE", 0.0), "launches_sampled": len(cs.get("FETCH_SIZE", [])),
                      "hbm_bytes_per_launch": (2.0 * g.get("FETCH_SIZE", 0.0) + g.get("WRITE_SIZE", 0.0)) * 1024.0}
    if "SQ_INSTS_VALU" not in g:
        continue
    cyc = g.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # summed over the 8 XCDs
    insts = g["SQ_INSTS_VALU"]
    d = {"launches_sampled": len(cs["SQ_INSTS_VALU"]), "gpu_cycles_per_launch": cyc, "valu_wave_instructions": insts,
         "salu_wave_instructions": g.get("SQ_INSTS_SALU"), "lds_wave_instructions": g.get("SQ_INSTS_LDS"),
         # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles: the issue peak is SIMDS * cycles / 2 wave-instructions per launch
         "valu_issue_fraction_2cycle": insts * 2.0 / (SIMDS * cyc) if cyc else None,
         # SQ_ACTIVE_INST_VALU counts, per wave and in quad-cycles, the time a VALU instruction of that wave is in flight (4 cycles each when a
         # wave issues alone): summed over waves it can exceed the SIMD's wall time, so *4/SIMDS/cycles is an occupancy-like figure, not a bound
         "valu_active_quadcycles_x4_per_simd_cycle": g.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (SIMDS * cyc) if cyc else None,
         "active_quadcycles_per_valu_instruction": g.get("SQ_ACTIVE_INST_VALU", 0.0) / insts if insts else None,
         "lane_utilisation": g.get("SQ_THREAD_CYCLES_VALU", 0.0) / (g.get("SQ_ACTIVE_INST_VALU", 1.0) * 64.0),
         "wait_any_fraction": g.get("SQ_WAIT_ANY", 0.0) / g.get("SQ_WAVE_CYCLES", 1.0), "wait_inst_fraction": g.get("SQ_WAIT_INST_ANY", 0.0) / g.get("SQ_WAVE_CYCLES", 1.0),
         "wait_inst_lds_fraction": (g["SQ_WAIT_INST_LDS"] / g["SQ_WAVE_CYCLES"]) if "SQ_WAIT_INST_LDS" in g else None,
         "lds_active_quadcycles_x4_per_cu_cycle": (g["SQ_ACTIVE_INST_LDS"] * 4.0 / (CUS * cyc)) if ("SQ_ACTIVE_INST_LDS" in g and cyc) else None,
         "lds_bank_conflict_cycles_per_cu_cycle": (g["SQ_LDS_BANK_CONFLICT"] / (CUS * cyc)) if ("SQ_LDS_BANK_CONFLICT" in g and cyc) else None,
         "waves": g.get("SQ_WAVES"), "busy_cycles": g.get("SQ_BUSY_CYCLES"),
         "raw": {c: g[c] for c in sorted(g) if c not in ("FETCH_SIZE", "WRITE_SIZE")}}
    if dur.get(k):
        d["avg_launch_ms_under_pmc"] = sum(dur[k]) / len(dur[k])
    valu[k] = d
    lines.append(f"{k}: launches={d['launches_sampled']} cycles={cyc:.3g} VALU={insts:.3g} SALU={g.get('SQ_INSTS_SALU', 0):.3g} LDS={g.get('SQ_INSTS_LDS', 0):.3g} "
                 f"issue2c={d['valu_issue_fraction_2cycle'] or 0:.2f} act4={d['valu_active_quadcycles_x4_per_simd_cycle'] or 0:.2f} qc/inst={d['active_quadcycles_per_valu_instruction'] or 0:.2f} "
                 f"lanes={d['lane_utilisation']:.2f} wait_any={d['wait_any_fraction']:.2f} wait_inst={d['wait_inst_fraction']:.2f} "
                 f"wait_lds={d['wait_inst_lds_fraction'] if d['wait_inst_lds_fraction'] is not None else -1:.2f} lds_act={d['lds_active_quadcycles_x4_per_cu_cycle'] or 0:.2f} "
                 f"lds_conf={d['lds_bank_conflict_cycles_per_cu_cycle'] or 0:.3f}")
for k, t in sorted(traffic.items()):
    lines.append(f"{k}: HBM bytes/launch {t['hbm_bytes_per_launch'] / 1e9:.3f} GB (fetch raw {t['fetch_size_kb_raw'] / 1e6:.3f} GB x2, write {t['write_size_kb'] / 1e6:.3f} GB)")
json.dump(valu, open(f"{out}/pmc_valu_config{config}.json", "w"), indent=1)
json.dump(traffic, open(f"{out}/pmc_traffic_config{config}.json", "w"), indent=1)
open(f"{out}/pmc_config{config}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
