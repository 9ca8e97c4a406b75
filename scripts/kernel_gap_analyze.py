import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "wf_" in r["Kernel_Name"]]
# find frames: from wf_raygen to wf_resolve
frames = []; cur = None
for r in rows:
    n = re.search(r"wf_[a-z_]+", r["Kernel_Name"]).group(0)
    if n == "wf_raygen": cur = []
    if cur is not None: cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if n == "wf_resolve" and cur: frames.append(cur); cur = None
frames = [f for f in frames if len(f) == 14][-20:]
tot = []; busy = []
for fr in frames:
    t0 = fr[0][1]; t1 = fr[-1][2]
    # union of intervals
    iv = sorted((s, e) for _, s, e in fr); u = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    tot.append((t1 - t0) / 1e3); busy.append(u / 1e3)
import statistics
print("frames", len(frames), "span %.1f us, kernels busy (union) %.1f us, gaps %.1f us" % (statistics.median(tot), statistics.median(busy), statistics.median(tot) - statistics.median(busy)))
dur = collections.defaultdict(list)
for fr in frames:
    for n, s, e in fr: dur[n].append((e - s) / 1e3)
print({k: round(statistics.median(v), 1) for k, v in dur.items()})
