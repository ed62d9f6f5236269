"""Build libvsel.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m visionselector_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvsel.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


# The hand-scheduled tile loops are GENERATED gfx950 inline-asm bodies (30.9 k lines for five kernels): they are produced here, at
# build time, from the generators under tools/ -- history tracks the generators and csrc/generated_bodies.sha256 (tests/test_*_generated_cpu.py
# hold the generators to those hashes), not the 31 k lines.  (generator, environment prefix, options, macro prefix, output)
ROOT_DIR = os.path.dirname(PKG_DIR)
GENERATED = (
    ("gen_attn_fwd64.py", "F64", "", "", "attn_fwd64_body.inc"),
    ("gen_attn_fwd64.py", "F64", "raw=1", "VSEL_FWD64P", "attn_fwd64_parts_body.inc"),
    ("gen_attn_fwd64.py", "F64", "heads=1,xitem=1,epi=1,qearly=1", "VSEL_GQA64", "attn_fwd_gqa64_body.inc"),
    ("gen_attn_bwd_dq64.py", "DQ64", "", "", "attn_bwd_dq64_body.inc"),
    ("gen_attn_bwd_dkdv64.py", "DKDV64", "", "", "attn_bwd_dkdv64_body.inc"),
)
HASHES_PATH = os.path.join(CSRC, "generated_bodies.sha256")


def generate_body(entry, out_path: str) -> str:
    """Run one generator into out_path; returns the text."""
    gen, env_prefix, opts, macro_prefix, _ = entry
    env = dict(os.environ)
    env[f"{env_prefix}_OUT"] = out_path
    env[f"{env_prefix}_OPTS"] = opts
    if macro_prefix:
        env[f"{env_prefix}_PREFIX"] = macro_prefix
    else:
        env.pop(f"{env_prefix}_PREFIX", None)
    subprocess.check_call([sys.executable, os.path.join(ROOT_DIR, "tools", gen)], env=env, stdout=subprocess.DEVNULL)
    with open(out_path) as f:
        return f.read()


def recorded_hashes() -> dict:
    with open(HASHES_PATH) as f:
        return dict(reversed(ln.split()) for ln in f if ln.strip() and not ln.startswith("#"))


def generate_bodies(verbose: bool = False) -> None:
    """csrc/*_body.inc from the generators (rewritten only when the text changes, so object files stay fresh); every body must match
    the hash history holds for it -- a generator edit is committed together with `python -m visionselector_amd.build --rehash`."""
    import hashlib
    import tempfile
    want = recorded_hashes()
    for entry in GENERATED:
        dst = os.path.join(CSRC, entry[4])
        with tempfile.TemporaryDirectory() as td:
            text = generate_body(entry, os.path.join(td, "body.inc"))
        got = hashlib.sha256(text.encode()).hexdigest()
        if want.get(entry[4]) != got:
            raise RuntimeError(f"{entry[4]}: tools/{entry[0]} writes sha256 {got}, csrc/generated_bodies.sha256 records {want.get(entry[4])}; "
                               "if the generator was changed on purpose run `python -m visionselector_amd.build --rehash`")
        old = open(dst).read() if os.path.exists(dst) else None
        if old != text:
            with open(dst, "w") as f:
                f.write(text)
            if verbose:
                print(f"[vsel build] generated {entry[4]} ({text.count(chr(10))} lines)", flush=True)


def rehash() -> None:
    import hashlib
    import tempfile
    lines = ["# sha256 of the generated kernel bodies (visionselector_amd/build.py GENERATED): hash  file\n"]
    for entry in GENERATED:
        with tempfile.TemporaryDirectory() as td:
            text = generate_body(entry, os.path.join(td, "body.inc"))
        lines.append(f"{hashlib.sha256(text.encode()).hexdigest()}  {entry[4]}\n")
    with open(HASHES_PATH, "w") as f:
        f.writelines(lines)


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if any(not os.path.exists(os.path.join(CSRC, e[4])) for e in GENERATED):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        [os.path.join(os.path.dirname(PKG_DIR), "include", "vsel.h")]          # (*.inc: the generated kernel bodies)
    return any(os.path.getmtime(p) > t for p in deps)


# Kernels that issue LDS reads from inline asm and wait for them later (csrc/attn_common.h): a register copy or spill between
# issue and wait would read stale data, so these must compile without spilled VGPRs and without scratch.
# (source file, mangled-name fragment).  The 8-wave dK/dV kernel (both instantiations) is on the list since round 4: its output
# epilogue converts with v_cvt_pk_bf16_f32 instead of the software rounding whose constants pushed it to 256 + 25 registers.
ASM_READ_KERNELS = (("attn.hip", "varlen_attn_fwd_kernelILb1E"), ("attn_bwd.hip", "attn_bwd_dkdv2_kernelILb0E"),
                    ("attn_bwd.hip", "attn_bwd_dkdv2_kernelILb1E"), ("attn_bwd.hip", "attn_bwd_dq_kernel"),
                    ("attn_fwd_gqa.hip", "attn_fwd_gqa_kernel"),
                    # the generated one-statement bodies clobber v32-255 / a0-255 / s40-99: everything the C++ around them keeps live must
                    # fit in what is left, without scratch
                    ("attn_fwd64.hip", "attn_fwd64_kernel"), ("attn_fwd_gqa64.hip", "attn_fwd_gqa64_kernel"), ("attn_bwd_dq64.hip", "attn_bwd_dq64_kernel"),
                    ("attn_bwd_dkdv64.hip", "attn_bwd_dkdv64_kernel"), ("attn_fwd64_parts.hip", "attn_fwd64_parts_kernel"))


def extra_flags() -> list:
    """VSEL_HIPCC_FLAGS (e.g. -DVSEL_TRACE, -save-temps).  Optimisation-level and debug-info flags are refused: the attention
    kernels' asm-issued LDS reads rely on the -O3 register allocation (no spills, checked below)."""
    flags = os.environ.get("VSEL_HIPCC_FLAGS", "").split()
    bad = [f for f in flags if f.startswith("-O") or f.startswith("-g") or f in ("-fno-inline", "-fno-unroll-loops") or
           f.startswith("-DVSEL_EXPERIMENT") or "_KO_" in f]          # (knock-out experiments compute WRONG results: tools/build_variant.sh only)
    if bad:
        raise ValueError(f"VSEL_HIPCC_FLAGS: {bad} would change code generation of the hand-scheduled kernels; not accepted")
    return flags


def check_no_spills(src: str, remarks: str) -> None:
    """Parse -Rpass-analysis=kernel-resource-usage remarks: every ASM_READ_KERNELS instantiation needs 0 spilled VGPRs."""
    import re
    frags = [f for s_, f in ASM_READ_KERNELS if os.path.basename(src) == s_]
    if not frags:
        return
    seen = set()
    for block in remarks.split("Function Name: ")[1:]:
        name = block.split()[0]
        hit = [f for f in frags if f in name]
        if not hit:
            continue
        m = re.search(r"VGPRs Spill: (\d+)", block)
        if m is None:
            raise RuntimeError(f"no resource-usage remark for {name}")
        seen.update(hit)
        sc = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", block)
        if sc is not None and int(sc.group(1)) != 0:
            raise RuntimeError(f"{name}: {sc.group(1)} bytes of scratch per lane -- a tile-loop lambda was not inlined or an array was "
                               "indexed dynamically; the hand-scheduled kernels must keep everything in registers")
        if int(m.group(1)) != 0:
            raise RuntimeError(f"{name}: {m.group(1)} spilled VGPRs -- its asm-issued LDS reads (attn_common.h) would read stale "
                               "registers; reduce live registers before shipping this build")
    missing = [f for f in frags if f not in seen]
    if missing:
        raise RuntimeError(f"{src}: no kernel matched {missing}; update build.ASM_READ_KERNELS")


def build_native(force: bool = False, verbose: bool = True) -> str:
    """Compile every csrc/*.hip into libvsel.so.  hipcc cross-compiles gfx950 without a GPU."""
    if not force and not _stale():
        return LIB_PATH
    generate_bodies(verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    build_dir = os.path.join(PKG_DIR, "build")
    os.makedirs(build_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(build_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(p) for p in [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))
                + [os.path.join(os.path.dirname(PKG_DIR), "include", "vsel.h")]):
            continue
        checked = any(os.path.basename(src) == s_ for s_, _ in ASM_READ_KERNELS)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + extra_flags() + \
            (["-Rpass-analysis=kernel-resource-usage"] if checked else []) + ["-c", src, "-o", obj]
        if verbose:
            print("[vsel build]", " ".join(cmd), flush=True)
        procs.append((cmd, src, obj, checked, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, src, obj, checked, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed ({p.returncode}): {' '.join(cmd)}\n{out}")
        if checked:
            try:
                check_no_spills(src, out)
            except RuntimeError:
                os.remove(obj)                      # never link an object that failed the check
                raise
            out = "\n".join(ln for ln in out.splitlines() if "kernel-resource-usage" not in ln and "remark:" not in ln)
            if "warning:" not in out and "error:" not in out:
                out = ""                            # only the source-context lines of the remarks were left
        if verbose and out.strip():
            print(out)
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[vsel build]", " ".join(link), flush=True)
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed: {r.stdout}")
    return LIB_PATH


if __name__ == "__main__":
    if "--rehash" in sys.argv:
        rehash()
    print(build_native(force="--force" in sys.argv))
