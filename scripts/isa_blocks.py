"""Per-basic-block instruction summary of one kernel of a HIP source (instructions, VALU, LDS reads / writes, global loads / stores,
scratch, waits, closing branch): shows at a glance whether a hot loop is clean (no scratch, how many waits) and where a kernel issues its
memory operations one round trip at a time.  Found the six-round-trip hypothesis scan in front of K3's plane loop (profiles/r06_experiments.md).
Usage: python scripts/isa_blocks.py warp_lds.hip <substring of the mangled kernel name> [min instructions per block, default 25] [-DFOO ...]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cds_mvsnet_amd", "csrc")
src, pat = sys.argv[1], sys.argv[2]
rest = sys.argv[3:]
nmin = int(rest.pop(0)) if rest and rest[0].isdigit() else 25
asm = "/tmp/_isa_blocks.s"
flags = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]        # the library's build (Makefile NOPK)
if src in ("feat_cl.hip", "conv2d_sbf.hip"):
    flags.append("-fno-slp-vectorize")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *flags, *rest, "-S",
                "--cuda-device-only", src, "-o", asm], cwd=csrc, check=True, stderr=subprocess.DEVNULL)
txt = open(asm).read().split("\n")
starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\S*:", l) and pat in l]
if not starts:
    sys.exit(f"no kernel label contains {pat!r}")
for st in starts:
    lines = txt[st:]
    lines = lines[:[i for i, l in enumerate(lines) if "s_endpgm" in l][0] + 1]
    print(subprocess.run(["c++filt", txt[st].split(":")[0]], capture_output=True, text=True).stdout.strip()[:160])
    blocks, cur = [], ("entry", 0, [])
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), i, [])
        else:
            cur[2].append(l)
    blocks.append(cur)
    for name, at, body in blocks:
        ins = [b.strip().split()[0] for b in body if b.strip() and not b.strip().startswith((";", "."))]
        f = lambda p: sum(1 for x in ins if x.startswith(p))
        br = [b.strip() for b in body if "s_cbranch" in b or "s_branch" in b]
        if len(ins) >= nmin:
            print(f"  {name:12s} +{at:5d} n={len(ins):5d} valu={f('v_'):5d} mfma={f('v_mfma'):4d} ds_rd={f('ds_read'):3d} ds_wr={f('ds_write'):3d} "
                  f"gld={f('global_load') + f('buffer_load'):3d} gst={f('global_store') + f('buffer_store'):3d} scr={f('scratch_'):3d} "
                  f"wait={f('s_waitcnt'):3d} bar={f('s_barrier'):2d}  {br[-1] if br else ''}")
