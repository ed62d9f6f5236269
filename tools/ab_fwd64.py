#!/usr/bin/env python3
"""Same-box A/B of schedule variants of the 64-rows-per-wave attention kernels (generator options of tools/gen_attn_fwd64.py, or of
tools/gen_attn_bwd_dq64.py with --dq64, of tools/gen_attn_bwd_dkdv64.py with --dkdv64).

    python tools/ab_fwd64.py build [--dq64|--dkdv64] name1:key=val,key=val name2:...   # here (no GPU): one libvsel_<name>.so per option set
    python tools/ab_fwd64.py run [--dq64|--dkdv64] [rounds] [--shapes 16x4096,4x8192]  # on the GPU box: alternating runs; forward TFLOP/s per
                                                                              # variant, or (--dq64 / --dkdv64) microseconds of that pass's kernel

A variant = the shipped library with csrc/attn_fwd64.hip recompiled against another generated body (visionselector_amd/build/variants/)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "visionselector_amd", "build", "variants")
PKG = os.path.join(ROOT, "visionselector_amd")


BWD = next((m for m in ("dq64", "dkdv64") if "--" + m in sys.argv), None)      # which backward pass's variants (None: the forward's)
if BWD:
    sys.argv.remove("--" + BWD)
DQ64 = BWD is not None
GEN, SRC, OBJ_SKIP, ENV_OPTS, ENV_OUT, DEF = {
    None: ("gen_attn_fwd64.py", "attn_fwd64.hip", "attn_fwd64.o", "F64_OPTS", "F64_OUT", "VSEL_FWD64_BODY"),
    "dq64": ("gen_attn_bwd_dq64.py", "attn_bwd_dq64.hip", "attn_bwd_dq64.o", "DQ64_OPTS", "DQ64_OUT", "VSEL_DQ64_BODY"),
    "dkdv64": ("gen_attn_bwd_dkdv64.py", "attn_bwd_dkdv64.hip", "attn_bwd_dkdv64.o", "DKDV64_OPTS", "DKDV64_OUT", "VSEL_DKDV64_BODY")}[BWD]
BWD_ENV, BWD_NEW, BWD_OLD = {None: ("", "", ""), "dq64": ("VSEL_ATTN_BWD_DQ64", "attn_bwd_dq64_kernel", "attn_bwd_dq_kernel"),
                             "dkdv64": ("VSEL_ATTN_BWD_DKDV64", "attn_bwd_dkdv64_kernel", "attn_bwd_dkdv_kernel")}[BWD]


def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for f in glob.glob(os.path.join(VDIR, "*")):
        os.remove(f)
    subprocess.check_call([sys.executable, "-m", "visionselector_amd.build"], cwd=ROOT)
    objs = [o for o in glob.glob(os.path.join(PKG, "build", "*.o")) if os.path.basename(o) != OBJ_SKIP]
    procs = []
    for spec in specs:
        name, _, opts = spec.partition(":")
        inc = os.path.join(VDIR, name + ".inc")
        env = dict(os.environ, **{ENV_OPTS: opts, ENV_OUT: inc})
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", GEN)], env=env, stdout=subprocess.DEVNULL)
        obj = os.path.join(VDIR, name + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f'-D{DEF}="{inc}"'] + (["-DVSEL_FWD64_TRACE"] if "trace=1" in opts else []) + ["-c",
               os.path.join(PKG, "csrc", SRC), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        lib = os.path.join(VDIR, f"libvsel_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + objs)
        os.remove(obj)
        print("built", lib)


def bench_one(lib, shapes, rows64=-1):
    code = f"""
import sys, json, os, torch
os.environ["VSEL_ATTN_ROWS64"] = "{rows64}"
sys.path.insert(0, {ROOT!r})
from visionselector_amd import _native
_native.LIB_PATH = {lib!r}
from visionselector_amd import ops
out = {{}}
for nseq, L in {shapes!r}:
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    for _ in range(60):          # (a fresh process needs ~100 ms of load before the clocks settle: 3 warm-up calls read 10 % low)
        ops.varlen_attn(q, k, v, cu, L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.varlen_attn(q, k, v, cu, L)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out[f"{{nseq}}x{{L}}"] = round(4.0 * L * L * 28 * 128 / 2 * nseq / (ms * 1e-3) / 1e12, 1)
print("RESULT " + json.dumps(out))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    for ln in r.stdout.splitlines():
        if ln.startswith("RESULT "):
            return json.loads(ln[7:])
    return {"error": (r.stderr or r.stdout)[-300:]}


def bench_dq64(lib, shapes, dq64=1):
    code = f"""
import sys, json, os, torch
os.environ["{BWD_ENV}"] = "{dq64}"
sys.path.insert(0, {ROOT!r})
from visionselector_amd import _native
_native.LIB_PATH = {lib!r}
from visionselector_amd import ops
out = {{}}
for nseq, L in {shapes!r}:
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    o, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    for _ in range(12):
        ops.varlen_attn_bwd(do, q, k, v, o, lse, cu, L)
    _native.profile_start()
    for _ in range(8):
        ops.varlen_attn_bwd(do, q, k, v, o, lse, cu, L)
    prof = _native.profile_stop()
    name = "{BWD_NEW}" if {dq64} else "{BWD_OLD}"
    out[f"{{nseq}}x{{L}}"] = round(prof[name][0] / prof[name][1] * 1e3, 1)
print("RESULT " + json.dumps(out))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    for ln in r.stdout.splitlines():
        if ln.startswith("RESULT "):
            return json.loads(ln[7:])
    return {"error": (r.stderr or r.stdout)[-300:]}


def run(rounds, shapes):
    if DQ64:
        libs = sorted(glob.glob(os.path.join(VDIR, "libvsel_*.so")))
        table = {}
        for r in range(rounds):
            for lib in [None] + libs:
                name = os.path.basename(lib)[8:-3] if lib else BWD_OLD
                res = bench_dq64(lib, shapes) if lib else bench_dq64(os.path.join(PKG, "libvsel.so"), shapes, dq64=0)
                for k, val in res.items():
                    table.setdefault(name, {}).setdefault(k, []).append(val)
        for name, row in table.items():
            print(json.dumps({"variant": name, **row}), flush=True)
        return
    libs = sorted(glob.glob(os.path.join(VDIR, "libvsel_*.so")))
    table = {}
    for r in range(rounds):
        for lib in [None] + libs:
            name = os.path.basename(lib)[8:-3] if lib else "rows32"
            res = bench_one(lib, shapes) if lib else bench_one(os.path.join(PKG, "libvsel.so"), shapes, rows64=0)
            for k, val in res.items():
                table.setdefault(name, {}).setdefault(k, []).append(val)
    for name, row in table.items():
        print(json.dumps({"variant": name, **row}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        rounds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
        shapes = [(16, 4096), (4, 8192)]
        if "--shapes" in sys.argv:
            shapes = [tuple(int(x) for x in sh.split("x")) for sh in sys.argv[sys.argv.index("--shapes") + 1].split(",")]
        run(rounds, shapes)
