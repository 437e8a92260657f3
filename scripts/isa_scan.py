"""Static scan of the gfx950 ISA of every kernel for two code-generation accidents that cost this project whole factors in round 5
(profiles/r05_experiments.md): (1) staging loops the compiler left as ONE global load followed by s_waitcnt vmcnt(0) per iteration
(every element pays a full memory round trip), (2) MFMAs followed by s_nop + v_accvgpr_read (the accumulators copied out of the AGPRs
around every MFMA because of a branch in the K loop).  Usage: python scripts/isa_scan.py  (needs hipcc; no GPU)."""
import glob, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "cds_mvsnet_amd", "csrc")
out = "/tmp/cds_isa"
os.makedirs(out, exist_ok=True)
procs = []
for f in sorted(glob.glob(os.path.join(src, "*.hip"))):
    extra = ["-fno-slp-vectorize"] if os.path.basename(f) in ("feat_cl.hip", "conv2d_sbf.hip") else []
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra, "-S",
                                   "--cuda-device-only", f, "-o", os.path.join(out, os.path.basename(f) + ".s")], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()


def demangle(n):
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        d = n
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    return re.sub(r"\(.*$", "", d)[:84]


flagged = 0
for f in sorted(glob.glob(os.path.join(out, "*.s"))):
    name, body = None, []
    for line in open(f):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(line)
        if "s_endpgm" in line:
            # (1) inside loop bodies (between a label and a backward branch to it): loads and full waits
            serial = 0
            labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
            for i, l in enumerate(body):
                m = re.search(r"s_cbranch_\w+ (\.LBB\w+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < i:
                    loop = body[labels[m.group(1)]:i]
                    loads = sum(1 for x in loop if re.search(r"\b(global|buffer|flat)_load", x))
                    waits = sum(1 for x in loop if "s_waitcnt vmcnt(0)" in x)
                    inner = sum(1 for x in loop if re.match(r"^\.LBB", x))
                    if inner <= 1 and 1 <= loads <= 2 and waits >= loads and len(loop) < 120:
                        serial += 1
            stall = 0
            for i, l in enumerate(body):
                if "v_mfma" in l and any("s_nop" in x for x in body[i + 1:i + 3]) and any("accvgpr_read" in x for x in body[i + 1:i + 6]):
                    stall += 1
            nm = sum(1 for l in body if "v_mfma" in l)
            if serial or stall >= 2:
                flagged += 1
                print(f"{os.path.basename(f)[:-2]:18s} {demangle(name):84s} serial-load loops={serial} mfma={nm} mfma+nop+accread={stall}")
            name = None
print(f"{flagged} kernels flagged")
