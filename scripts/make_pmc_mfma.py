"""MFMA-pipe utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES`
summary (scripts/pmc_summary.py output).  GRBM_GUI_ACTIVE is summed over the 8 XCDs; utilisation = MFMA busy cycles /
(GUI_ACTIVE / 8 x 1024 SIMDs).  Usage: make_pmc_mfma.py <summary.txt> <out.md>"""
import re, sys
rows, cur = [], None
for ln in open(sys.argv[1]):
    if not ln.startswith(" "):
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", ln.strip()).replace("void ", "").split("(")[0]}
        rows.append(cur)
    else:
        m = re.match(r"\s+(\S+)\s+mean (\S+)\s+\(n=(\d+)\)", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2)); cur["n"] = int(m.group(3))
out = ["# MFMA-pipe utilisation per kernel (PMC), round 1", "",
       "Command: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python scripts/time_forward.py 1184 1600 5`",
       "(full cascade forward, 1600x1184, N=5; counter pass only, no trace domains; means over all launches of a kernel, i.e.",
       "over the three cascade stages).  `GRBM_GUI_ACTIVE` is summed over the 8 XCDs; a v_mfma_f32_16x16x4_f32 occupies its",
       "SIMD's matrix pipe for 32 cycles.  utilisation = MFMA busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs).", "",
       "| kernel | launches | GUI_ACTIVE / 8 (cycles) | MFMA busy cycles | MFMA pipe utilisation |", "|---|---|---|---|---|"]
for r in rows:
    if r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and "GRBM_GUI_ACTIVE" in r:
        cyc = r["GRBM_GUI_ACTIVE"] / 8
        out.append(f"| `{r['name']}` | {r['n']} | {cyc:.3g} | {r['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} | {100 * r['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.0f} % |")
out += ["", "Kernels without MFMA instructions (the VALU convolutions, warps, blends) report 0 busy cycles and are omitted.",
        "Cross-check for `conv2d_k3_c16_mfma_kernel`: 36 MFMAs per 16 pixels x (7.58 + 1.89 + 0.47) M pixels of the three stages",
        "x 32 cycles / 3 launches = 2.39e8 busy cycles per launch on average, the counter reads 2.42e8."]
open(sys.argv[2], "w").write("\n".join(out) + "\n")
