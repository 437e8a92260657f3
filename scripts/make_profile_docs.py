"""Turn the outputs of scripts/ab/r05_profiles.sh (gpurun_out/<tag>/) into the committed profiles/<tag>_* documents."""
import json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")


def pmc_rows(path):
    rows, lines, i = [], open(path).read().split("\n"), 0
    while i < len(lines):
        if lines[i] and not lines[i].startswith(" "):
            name, d, j = lines[i], {}, i + 1
            while j < len(lines) and lines[j].startswith(" "):
                m = re.match(r"\s+(\S+)\s+mean (\S+)\s+\(n=(\d+)\)", lines[j])
                if m:
                    d[m.group(1)] = (float(m.group(2)), int(m.group(3)))
                j += 1
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0] > 0:
                gui, mf = d["GRBM_GUI_ACTIVE"][0] / 8, d["SQ_VALU_MFMA_BUSY_CYCLES"][0]
                short = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
                short = re.sub(r"\(float const\*.*$|\(HIP_vector.*$", "", short)[:70]
                rows.append((short, d["GRBM_GUI_ACTIVE"][1], gui, mf, 100.0 * mf / (gui * 1024)))
            i = j
        else:
            i += 1
    return rows


def table(rows):
    out = ["| kernel | launches | GUI_ACTIVE / 8 (cycles) | MFMA busy cycles | MFMA pipe utilisation |", "|---|---|---|---|---|"]
    out += [f"| `{n}` | {c} | {g:.3g} | {m:.4g} | {u:.0f} % |" for n, c, g, m, u in rows]
    return "\n".join(out)


shutil.copy(os.path.join(src, f"{tag}_kernel_stats.md"), os.path.join(dst, f"{tag}_kernel_stats.md"))
if os.path.exists(os.path.join(src, "pmc_traffic.md")):
    shutil.copy(os.path.join(src, "pmc_traffic.md"), os.path.join(dst, f"{tag}_pmc_hbm_traffic.md"))
with open(os.path.join(dst, f"{tag}_pmc_mfma_utilisation.md"), "w") as f:
    f.write(f"# MFMA-pipe utilisation per kernel (PMC), round {tag[1:]}\n\n"
            "Command: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python bench.py --no-extras --steps 5 --cpu-sample 0`\n"
            "(M1 headline workload: 640x512, D=192, C=8, N=5; counter pass only, no trace domains; means over all launches of a kernel).\n"
            "`GRBM_GUI_ACTIVE` is summed over the 8 XCDs; a v_mfma_f32_16x16x32_bf16 occupies its SIMD's matrix pipe for 16 cycles.\n"
            "utilisation = MFMA busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs).\n\n")
    f.write(table(pmc_rows(os.path.join(src, f"{tag}_pmc_mfma.txt"))) + "\n\n")
    f.write("## The 1600x1184 cascade forward (same counters over `python scripts/time_forward.py 1184 1600 5`, 7 forwards)\n\n")
    f.write(table(pmc_rows(os.path.join(src, f"{tag}_m3_pmc_mfma.txt"))) + "\n\n")
    notes = os.path.join(dst, f"{tag}_pmc_mfma_notes.md")
    if os.path.exists(notes):
        f.write(open(notes).read())
with open(os.path.join(dst, f"{tag}_m3_cascade_breakdown.md"), "w") as f:
    f.write(f"# 1600x1184 cascade forward (BASELINE config 3, N=5): kernel time by kernel over 7 forwards, round {tag[1:]}\n\n"
            "Command: `rocprofv3 --kernel-trace --stats -- python scripts/time_forward.py 1184 1600 5` (2 warm-up + 5 timed forwards; totals over\n"
            "all 7; the `at::native::*` and `__amd_rocclr_copyBuffer` rows are the one-time model set-up (parameter upload and weight packing) of\n"
            "the script, not the forward: `scripts/ab/r05_aten_ops.py` counts ONE ATen launch on device tensors inside a forward, the image stack).\n\n```\n")
    f.write(open(os.path.join(src, f"{tag}_m3_breakdown.txt")).read())
    f.write("```\n\n")
    notes = os.path.join(dst, f"{tag}_m3_notes.md")
    if os.path.exists(notes):
        f.write(open(notes).read())
line = [l for l in open(os.path.join(src, "bench_b.json")).read().split("\n") if l.startswith("{")][-1]
json.loads(line)
open(os.path.join(dst, f"{tag}_bench_line.json"), "w").write(line + "\n")
print("profiles written for", tag)
