"""Static checks on the gfx950 code of the Cholesky kernels (no GPU needed; ~40 s of hipcc):
  * no scratch access inside a K loop (a block with the operand loads of one K slab and its 32 MFMAs): hipcc places a
    spill reload there now and then -- its wait also drains the hand-counted operand prefetch (measured: 3 % of a
    dataflow launch);
  * the wide panel kernel has no scratch at all and stays at 128 VGPRs (4 waves per SIMD).
    python tools/check_isa.py            -> prints a summary, exit code 1 on a violation"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "starfish_amd", "csrc", "sf_chol.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernel_blocks(asm):
    """{kernel symbol: [(label, [instructions])]}"""
    out, cur, blk = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, blk = m.group(1), None
            out[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blk = (m.group(1), [])
            out[cur].append(blk)
            continue
        t = line.strip()
        if line.startswith("\t") and t and not t.startswith((";", ".")):
            if blk is None:
                blk = ("entry", [])
                out[cur].append(blk)
            blk[1].append(t.split(";")[0].strip())
            if t.startswith("s_endpgm"):
                cur = None
    return out


def check(verbose=True):
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "sf_chol.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", s, SRC,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        asm = open(s).read()
        remarks = r.stderr
    problems = []
    for name, blocks in kernel_blocks(asm).items():
        if not any(k in name for k in ("k_chol_panel", "k_potrf_dataflow")):
            continue
        for label, ins in blocks:
            mfma = sum(i.startswith("v_mfma") for i in ins)
            glds = sum("global_load_lds" in i for i in ins)
            scr = [i for i in ins if i.startswith("scratch_")]
            if mfma >= 32 and glds >= 3 and scr:
                problems.append(f"{name} {label}: {len(scr)} scratch access(es) inside a K loop: {scr[0]}")
    # resource usage of the wide kernel
    for m in re.finditer(r"Function Name: (\S*k_chol_panel_w\S*).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", remarks, re.S):
        if verbose:
            print(f"{m.group(1)}: {m.group(2)} VGPRs, {m.group(3)} B/lane scratch")
        if int(m.group(3)) != 0 or int(m.group(2)) > 128:
            problems.append(f"{m.group(1)}: {m.group(2)} VGPRs, {m.group(3)} B/lane of scratch (want <= 128, 0)")
    for m in re.finditer(r"Function Name: (\S*k_potrf_dataflow\S*).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", remarks, re.S):
        if verbose:
            print(f"{m.group(1)}: {m.group(2)} VGPRs, {m.group(3)} B/lane scratch (outside the K loops)")
    return problems


if __name__ == "__main__":
    p = check()
    for x in p:
        print("VIOLATION:", x)
    print("ok" if not p else f"{len(p)} violation(s)")
    sys.exit(1 if p else 0)
