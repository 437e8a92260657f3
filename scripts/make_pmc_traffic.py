"""Build profiles/pmc_traffic.json + a markdown table from the two PMC passes of scripts/run_k3_traffic.py.
Usage: make_pmc_traffic.py <fetch.db> <write.db> <out.md> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests as 64 B (calibrated here on the
kernels with known byte counts), so fetch bytes are doubled."""
import json, sqlite3, sys, collections

def means(db, counter):
    cur = sqlite3.connect(db).cursor()
    agg = collections.defaultdict(list)
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c == counter:
            agg[k].append(v)
    return {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}

fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
def pick(d, sub):
    for k, v in d.items():
        if sub in k:
            return v
    return float("nan")
h, w, D, C, N = 512, 640, 192, 8, 5
hw = h * w
known = {"volume_normalize": (4 * (C * D * hw + hw), 4 * C * D * hw), "chw_to_hwc": (4 * C * hw, 4 * C * hw),
         "vis_sum": (4 * (N - 1) * hw, 4 * hw)}
rows = [("warp_aggregate (K3)", "warp_aggregate", None), ("warp_entropy (K1)", "warp_entropy", None)] + \
       [(f"{k} (calibration)", k, v) for k, v in known.items()]
lines = ["| kernel | FETCH_SIZE raw (B) | WRITE_SIZE raw (B) | known read (B) | known write (B) | raw fetch / known | raw write / known |",
         "|---|---|---|---|---|---|---|"]
for label, sub, kn in rows:
    f, wr = pick(fetch, sub), pick(write, sub)
    if kn:
        lines.append(f"| {label} | {f:.0f} | {wr:.0f} | {kn[0]} | {kn[1]} | {f / kn[0]:.3f} | {wr / kn[1]:.3f} |")
    else:
        lines.append(f"| {label} | {f:.0f} | {wr:.0f} | - | - | - | - |")
b_alg = 4 * (C * D * hw + D * hw + 2 * (N - 1) * C * hw + (N - 1) * hw)
k3f, k3w = pick(fetch, "warp_aggregate"), pick(write, "warp_aggregate")
k1f, k1w = pick(fetch, "warp_entropy"), pick(write, "warp_entropy")
k3 = k3w + 2 * k3f
lines += ["", f"K3 warp_aggregate: HBM traffic per launch = WRITE + 2 x FETCH = {k3:.0f} B vs algorithmic {b_alg} B -> {k3 / b_alg:.3f}x.",
          f"K1 warp_entropy: {k1w + 2 * k1f:.0f} B."]
open(sys.argv[3], "w").write("\n".join(lines) + "\n")
json.dump({"M1": {"warp_aggregate_hbm_bytes": k3, "warp_aggregate_write_bytes": k3w,
                  "warp_aggregate_fetch_bytes_corrected": 2 * k3f, "warp_entropy_hbm_bytes": k1w + 2 * k1f,
                  "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units); FETCH_SIZE doubled "
                            "(gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section), calibrated on "
                            "volume_normalize / chw_to_hwc / vis_sum whose byte counts are known"}},
          open(sys.argv[4], "w"), indent=1)
print("\n".join(lines))
