"""The reference-side binding printed in INTEGRATION.md section 2 is executed AS WRITTEN.

A maintainer of the reference pastes that block in place of `F.relu(nn.Conv1d(...)(x))`
(phase3/archis/default.py:312-316), so it must track include/m2d.h: the CPU test checks
every `argtypes` list of the block against the header / the ctypes table, the GPU test
runs the block on cuda:0 and compares with torch's own conv in fp64.
"""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    hits = [b for b in blocks if "def conv1d_relu" in b]
    assert len(hits) == 1, "INTEGRATION.md must hold exactly one conv1d_relu binding stub"
    return hits[0]


def header_param_counts():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "m2d.h")).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(m2d_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


class _Recorder:
    """Stands in for ctypes.CDLL while the stub's binding lines run: records restype / argtypes."""

    class _Fn:
        restype = None
        argtypes = None

    def __init__(self):
        object.__setattr__(self, "fns", {})

    def __getattr__(self, name):
        return self.fns.setdefault(name, _Recorder._Fn())


def test_stub_argtypes_match_the_header():
    from music2dance_amd import _lib
    src = stub_source()
    head = src[:src.index("def conv1d_relu")]
    rec = _Recorder()
    env = {"ctypes": ctypes, "torch": torch}
    fake_ctypes = type("C", (), {k: getattr(ctypes, k) for k in dir(ctypes) if not k.startswith("__")})
    fake_ctypes.CDLL = staticmethod(lambda path: rec)
    env["ctypes"] = fake_ctypes
    exec(head.replace("import ctypes, torch", "pass"), env)
    counts = header_param_counts()
    assert "m2d_conv1d_fwd" in rec.fns
    for name, fn in rec.fns.items():
        assert name in counts, "%s is not declared in include/m2d.h" % name
        if fn.argtypes is not None:
            assert len(fn.argtypes) == counts[name], (name, len(fn.argtypes), counts[name])
            ref = _lib.SIGNATURES[name][1]
            # same C type class per position (pointer / int / float / size_t) as the package's own table
            for i, (a, b) in enumerate(zip(fn.argtypes, ref)):
                assert ctypes.sizeof(a) == ctypes.sizeof(b) and (a is ctypes.c_float) == (b is ctypes.c_float), (name, i)
    # the call itself passes one value per declared parameter
    call = re.search(r"lib\.m2d_conv1d_fwd\((.*?)\)\n\s*if rc", src, flags=re.S).group(1)
    depth, nargs = 0, 1
    for ch in call:
        depth += ch in "(["
        depth -= ch in ")]"
        nargs += (ch == "," and depth == 0)
    assert nargs == counts["m2d_conv1d_fwd"], (nargs, counts["m2d_conv1d_fwd"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 32, 4800, 64, 25, 4, 11),   # audio critic l2 (phase3/archis/default.py:299)
                                   (3, 128, 120, 128, 7, 1, 3),    # TemporalBlock conv (:201-204)
                                   (2, 5, 61, 9, 3, 2, 0)])        # ragged: chunk tails, no padding
def test_stub_runs_and_matches_torch_conv(shape):
    import torch.nn.functional as Fn
    B, Cin, L, Cout, k, stride, pad = shape
    cwd = os.getcwd()
    os.chdir(ROOT)  # the stub loads the library by its in-tree relative path
    try:
        env = {}
        exec(stub_source(), env)
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, L, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    y = env["conv1d_relu"](x, w, b, stride, pad)
    torch.cuda.synchronize()
    ref = Fn.relu(Fn.conv1d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=stride, padding=pad))
    assert y.shape == ref.shape
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-5
