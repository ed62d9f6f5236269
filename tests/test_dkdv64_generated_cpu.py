"""The per-item body of attn_bwd_dkdv64_kernel is GENERATED (tools/gen_attn_bwd_dkdv64.py -> csrc/attn_bwd_dkdv64_body.inc): the committed
file must be the committed generator's output, and every register the body names must be on the clobber list it declares."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionselector_amd import build as vbuild  # noqa: E402

vbuild.generate_bodies()                        # (what build() does first: the bodies are generated, not in history)
GEN = os.path.join(ROOT, "tools", "gen_attn_bwd_dkdv64.py")
INC = os.path.join(ROOT, "visionselector_amd", "csrc", "attn_bwd_dkdv64_body.inc")


def test_committed_body_is_the_generators_output(tmp_path):
    out = tmp_path / "body.inc"
    subprocess.check_call([sys.executable, GEN], env=dict(os.environ, DKDV64_OUT=str(out), DKDV64_OPTS=""), stdout=subprocess.DEVNULL)
    assert out.read_text() == open(INC).read(), "stale generated body: run `python -m visionselector_amd.build`"
    import hashlib
    assert hashlib.sha256(out.read_text().encode()).hexdigest() == vbuild.recorded_hashes()["attn_bwd_dkdv64_body.inc"], "generator changed: `python -m visionselector_amd.build --rehash`"


def test_body_register_budget_mfma_count_and_lds_size():
    text = open(INC).read()
    body, clob = text.split("#define VSEL_DKDV64_ASM_CLOBBERS")
    declared = set(re.findall(r'"([vas]\d+)"', clob))
    used = set()
    for kind, lo, hi in re.findall(r"\b([vas])\[(\d+):(\d+)\]", body):
        used.update(f"{kind}{i}" for i in range(int(lo), int(hi) + 1))
    used.update(re.findall(r"(?<![\w%\[])([vas]\d+)\b", body))
    assert used <= declared, sorted(used - declared)[:10]
    assert not {"s100", "s101"} & used                           # (left to the compiler: attn_bwd_dkdv64.hip keeps loop state around the body)
    # a step = SdP(t,0) dVdK(t-1,1) SdP(t,1) dVdK(t,0), 16 MFMAs each: steady / generic bodies x three ring slots, the first step (no
    # unit of a previous tile), and the drain of the last tile's second unit for each ring slot
    assert body.count("v_mfma_f32_32x32x16_bf16") == 6 * 64 + 48 + 3 * 16
    # Q[3] | dO[3] rings of 64 x 128 bf16 tiles, lse2[3][64] | D[3][64] floats
    assert int(re.search(r"#define VSEL_DKDV64_LDS_BYTES (\d+)", text).group(1)) == 6 * 64 * 128 * 2 + 2 * 3 * 64 * 4
