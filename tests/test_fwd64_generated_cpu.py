"""The per-item body of attn_fwd64_kernel is GENERATED at build time (tools/gen_attn_fwd64.py -> csrc/attn_fwd64_body.inc, git-ignored;
visionselector_amd/build.py).  History holds the generator and the sha256 of what it writes (csrc/generated_bodies.sha256): the
generator must reproduce that hash, the in-tree body the library was built from must be the generator's output, and the generator's
own consistency checks (counted LDS waits, register pipelines) must hold for the option sets the tools build."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionselector_amd import build as vbuild  # noqa: E402

vbuild.generate_bodies()                        # (what build() does first: the bodies are not in history)
GEN = os.path.join(ROOT, "tools", "gen_attn_fwd64.py")
INC = os.path.join(ROOT, "visionselector_amd", "csrc", "attn_fwd64_body.inc")
HEADS_OPTS = "heads=1,xitem=1,epi=1,qearly=1"          # options of the committed csrc/attn_fwd_gqa64_body.inc


def _gen(tmp_path, opts=""):
    out = tmp_path / "body.inc"
    env = dict(os.environ, F64_OUT=str(out), F64_OPTS=opts)
    subprocess.check_call([sys.executable, GEN], env=env, stdout=subprocess.DEVNULL)
    return out.read_text()


def _sha(text):
    return hashlib.sha256(text.encode()).hexdigest()


def test_committed_body_is_the_generators_output(tmp_path):
    text = _gen(tmp_path)
    assert text == open(INC).read(), "stale csrc/attn_fwd64_body.inc: run `python -m visionselector_amd.build`"
    assert _sha(text) == vbuild.recorded_hashes()["attn_fwd64_body.inc"], "generator changed: `python -m visionselector_amd.build --rehash`"


def test_every_generated_body_has_its_recorded_hash():
    want = vbuild.recorded_hashes()
    assert sorted(want) == sorted(e[4] for e in vbuild.GENERATED)
    for e in vbuild.GENERATED:
        assert _sha(open(os.path.join(vbuild.CSRC, e[4])).read()) == want[e[4]], e[4]


@pytest.mark.parametrize("opts", ["move_chunk=0", "pvsplit=1", "wgrp=0,srot=0", "dmak=top,dmav=top,pre=0", "trace=1",
                                  "early=57", "ko=fin+max+lds+dma"])
def test_generator_option_sets_are_consistent(tmp_path, opts):
    text = _gen(tmp_path, opts)
    assert text.count("v_mfma_f32_32x32x16_bf16") >= 352       # prologue S(0) + two steady, two generic, two last-tile step bodies


def test_body_register_budget():
    """every register the body names is inside the clobber list it declares (the compiler keeps v0..v31 and the low SGPRs)"""
    import re
    text = open(INC).read()
    body, clob = text.split("#define VSEL_FWD64_ASM_CLOBBERS")
    declared = set(re.findall(r'"([vas]\d+)"', clob))
    used = set()
    for kind, lo, hi in re.findall(r"\b([vas])\[(\d+):(\d+)\]", body):
        used.update(f"{kind}{i}" for i in range(int(lo), int(hi) + 1))
    used.update(re.findall(r"(?<![\w%\[])([vas]\d+)\b", body))
    assert used <= declared, sorted(used - declared)[:10]


def test_committed_parts_body_is_the_generators_output(tmp_path):
    """attn_fwd64_parts_body.inc = the same generator with raw = 1 (key-range parts: fp32 accumulators + (m, l) out, csrc/attn_fwd64_parts.hip)"""
    out = tmp_path / "body.inc"
    env = dict(os.environ, F64_OUT=str(out), F64_OPTS="raw=1", F64_PREFIX="VSEL_FWD64P")
    subprocess.check_call([sys.executable, GEN], env=env, stdout=subprocess.DEVNULL)
    inc = os.path.join(ROOT, "visionselector_amd", "csrc", "attn_fwd64_parts_body.inc")
    assert out.read_text() == open(inc).read(), "stale csrc/attn_fwd64_parts_body.inc: run `python -m visionselector_amd.build`"
    assert _sha(out.read_text()) == vbuild.recorded_hashes()["attn_fwd64_parts_body.inc"]


def test_committed_heads_body_is_the_generators_output(tmp_path):
    """attn_fwd_gqa64_body.inc = the same generator with heads = 1 (two q heads of a GQA group per wave) under its own macro prefix"""
    out = tmp_path / "body.inc"
    env = dict(os.environ, F64_OUT=str(out), F64_OPTS=HEADS_OPTS, F64_PREFIX="VSEL_GQA64")
    subprocess.check_call([sys.executable, GEN], env=env, stdout=subprocess.DEVNULL)
    inc = os.path.join(ROOT, "visionselector_amd", "csrc", "attn_fwd_gqa64_body.inc")
    assert out.read_text() == open(inc).read(), f"run F64_OPTS={HEADS_OPTS} F64_PREFIX=VSEL_GQA64 F64_OUT=... python tools/gen_attn_fwd64.py"
