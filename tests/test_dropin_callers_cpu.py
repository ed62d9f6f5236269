"""The drop-in import surface, checked against the reference's own callers.

Build container only: needs /root/reference (skipped on the GPU box).  For each script of the reference that imports
`compression_method.*` / `token_compression.*`, the import statements are extracted with `ast` and executed UNCHANGED with
`visionselector_amd/dropin` ahead of the reference's package root on sys.path:

  * every selector-path name (TransformerScorer, topk, *_forward_selector, *_Selector classes, replace_*) must resolve to
    this library;
  * every baseline module the same scripts import unconditionally (`token_compression.visionzip_official`,
    `.dynamic_model`, ...) must still be FOUND in the reference tree behind ours (dropin packages are namespace
    portions); a baseline that then fails inside the reference's own code under this image's transformers is the
    reference's incompatibility, not a shadowing defect, and is accepted only when raised from outside this repo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DROPIN = os.path.join(ROOT, "visionselector_amd", "dropin")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")

# (caller script, package root the reference's launch scripts put on PYTHONPATH for it)
CALLERS = [
    ("lmms-eval/lmms_eval/models/llava_onevision1_5_with_token_compression.py", "llava-ov-15"),       # :19-20, :113
    ("lmms-eval/lmms_eval/models/qwen2_5_vl_with_token_compression.py", "qwen-evaluation"),           # :20-23, :129
    ("lmms-eval/lmms_eval/models/qwen2_5_vl_with_token_compression_visual.py", "qwen-evaluation"),    # :16-17, :111
    ("qwen-evaluation/predict.py", "qwen-evaluation"),                                                # :75, :91-93
    ("llava-ov-15/predict_ov.py", "llava-ov-15"),                                                     # :70, :87
    ("qwen-vl-finetune/qwenvl/train/train_qwen_selector.py", "qwen-vl-finetune"),                     # :49-50
    ("llava-ov-15/src/train/train_sft_visionselector.py", "llava-ov-15"),                             # :14-15
]
# modules of the path this library provides; anything else under the two packages is a reference baseline
OURS = {"compression_method.selector_scorer", "compression_method.selector_model", "compression_method.modeling_selector",
        "compression_method.monkeypatch", "token_compression.selector_scorer", "token_compression.selector_model",
        "token_compression.monkeypatch"}


def _probe(caller, root):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_import_probe.py"), ROOT,
                        os.path.join(REF, caller), os.path.join(REF, root)],
                       capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE ")]
    assert lines, f"probe produced no result: rc={r.returncode}\n{r.stderr[-2000:]}"
    return json.loads(lines[-1][6:])


@pytest.mark.parametrize("caller,root", CALLERS, ids=[c[0].split("/")[-1] for c in CALLERS])
def test_reference_caller_imports_resolve(caller, root):
    rows = _probe(caller, root)
    assert rows, f"{caller}: no compression_method/token_compression imports found"
    for row in rows:
        where = f"{caller}:{row['line']}: {row['stmt']}"
        if row["module"] in OURS:
            assert row["ok"], f"{where} -> {row.get('error')}"
            assert row["origin"].startswith(DROPIN), f"{where} resolved to {row['origin']}"
            for name, mod in row["names"].items():
                # star imports also re-export helper modules (torch, ...); selector names must be ours
                # (`topk` is `TopK.apply`, a torch.autograd bound method: checked through TopK)
                if any(s in name for s in ("elector", "Scorer", "TopK", "replace_")):
                    assert mod.startswith(("visionselector_amd", "compression_method", "token_compression")), \
                        f"{where}: {name} comes from {mod}"
        else:
            assert row["origin"] and row["origin"].startswith(REF), \
                f"{where}: baseline module shadowed / not found ({row['origin']}, {row.get('find_error')})"
            if not row["ok"]:
                assert not row["raised_in"].startswith(ROOT), f"{where} failed inside this repo: {row['error']}"


def test_star_import_surface_of_ov_training_script():
    """`from compression_method.selector_model import *` (llava-ov-15/src/train/train_sft_visionselector.py:14-15) must
    bring in every name that script then uses (:219-225)."""
    rows = _probe("llava-ov-15/src/train/train_sft_visionselector.py", "llava-ov-15")
    names = {}
    for row in rows:
        names.update(row["names"])
    for need in ("TransformerScorer", "topk", "TopK", "llavaov15_vision_tower_forward_selector",
                 "llavaov15_vlmodel_forward_selector", "llavaov15_generation_forward_selector"):
        assert need in names, f"star import lost {need}"


def test_ov_selector_classes_are_built_on_the_reference_bases():
    """The three names of llava-ov-15/compression_method/modeling_selector.py:68,188,339 resolve lazily, once, to
    subclasses of the caller's own OV classes; the tower class constructs and carries the scorer's state-dict keys."""
    code = r"""
import sys
sys.dont_write_bytecode = True
sys.path[:0] = [%r, %r, %r]
import compression_method.modeling_selector as ms
assert 'llavaonevision1_5' not in sys.modules            # importing the shim does not need the OV code
from compression_method.modeling_selector import LLaVAOneVision1_5_ForConditionalGeneration_Selector as CG
from compression_method.modeling_selector import RiceTransformerPretrainedModel_Selector as Rice
import llavaonevision1_5.modeling_llavaonevision1_5 as ov
assert issubclass(CG, ov.LLaVAOneVision1_5_ForConditionalGeneration)
assert issubclass(ms.LLaVAOneVision1_5_Model_Selector, ov.LLaVAOneVision1_5_Model)
assert issubclass(Rice, ov.RiceTransformerPretrainedModel)
assert ms.LLaVAOneVision1_5_ForConditionalGeneration_Selector is CG          # built once
assert CG.__name__ == 'LLaVAOneVision1_5_ForConditionalGeneration_Selector'
try:
    ms.NoSuchThing
    raise SystemExit('missing attribute did not raise')
except AttributeError:
    pass
from llavaonevision1_5.configuration_llavaonevision1_5 import RiceConfig
t = Rice(RiceConfig(depth=1, embed_dim=32, hidden_size=32, intermediate_size=64, num_heads=2, text_hidden_size=64,
                    patch_size=14, spatial_merge_size=2))
keys = sorted(k for k in t.state_dict() if k.startswith('importance_scorer.'))
assert keys == ['importance_scorer.k_proj.bias', 'importance_scorer.k_proj.weight',
                'importance_scorer.q_proj.bias', 'importance_scorer.q_proj.weight'], keys
assert tuple(t.importance_scorer.q_proj.weight.shape) == (32, 64)            # Linear(D, D // 2), modeling_selector.py:101
assert t.budgets == 1.0
print('OV-OK')
""" % (DROPIN, ROOT, os.path.join(REF, "llava-ov-15"))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert "OV-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_baseline_method_is_forwarded_to_the_reference_monkeypatch():
    """`replace_qwen25vl(args, model, 'fastv')`: with the reference root behind dropin/, the reference's own branch
    (qwen-evaluation/token_compression/monkeypatch.py:71-76) runs; without it, NotImplementedError as before."""
    code = r"""
import sys, types, torch
sys.dont_write_bytecode = True
sys.path[:0] = [%r, %r, %r]
from token_compression.monkeypatch import replace_qwen25vl
sentinel = object()
assert replace_qwen25vl(None, sentinel, 'selector') is sentinel
replace_qwen25vl(types.SimpleNamespace(), torch.nn.Linear(2, 2), 'fastv')
ref = sys.modules['token_compression._reference_monkeypatch']
assert ref.__file__.startswith(%r), ref.__file__
assert sys.modules['token_compression.fastv'].__file__.startswith(%r)      # its relative imports resolved behind ours
print('FWD-OK')
""" % (DROPIN, ROOT, os.path.join(REF, "qwen-evaluation"), REF, REF)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert "using fastv" in r.stdout                     # the reference branch's own print (monkeypatch.py:72)
    assert "FWD-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
