"""The per-item body of attn_bwd_dq64_kernel is GENERATED (tools/gen_attn_bwd_dq64.py -> csrc/attn_bwd_dq64_body.inc): the committed file
must be the committed generator's output, and every register the body names must be on the clobber list it declares."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionselector_amd import build as vbuild  # noqa: E402

vbuild.generate_bodies()                        # (what build() does first: the bodies are generated, not in history)
GEN = os.path.join(ROOT, "tools", "gen_attn_bwd_dq64.py")
INC = os.path.join(ROOT, "visionselector_amd", "csrc", "attn_bwd_dq64_body.inc")


def test_committed_body_is_the_generators_output(tmp_path):
    out = tmp_path / "body.inc"
    subprocess.check_call([sys.executable, GEN], env=dict(os.environ, DQ64_OUT=str(out), DQ64_OPTS=""), stdout=subprocess.DEVNULL)
    assert out.read_text() == open(INC).read(), "stale generated body: run `python -m visionselector_amd.build`"
    import hashlib
    assert hashlib.sha256(out.read_text().encode()).hexdigest() == vbuild.recorded_hashes()["attn_bwd_dq64_body.inc"], "generator changed: `python -m visionselector_amd.build --rehash`"


def test_body_register_budget_and_mfma_count():
    text = open(INC).read()
    body, clob = text.split("#define VSEL_DQ64_ASM_CLOBBERS")
    declared = set(re.findall(r'"([vas]\d+)"', clob))
    used = set()
    for kind, lo, hi in re.findall(r"\b([vas])\[(\d+):(\d+)\]", body):
        used.update(f"{kind}{i}" for i in range(int(lo), int(hi) + 1))
    used.update(re.findall(r"(?<![\w%\[])([vas]\d+)\b", body))
    assert used <= declared, sorted(used - declared)[:10]
    # five step bodies of 96 - 8 .. 96 MFMAs (steady / generic x two ring slots, first step) + the drains
    assert body.count("v_mfma_f32_32x32x16_bf16") == 2 * 96 + 2 * 96 + 88 + 3 * 8
