"""The C-ABI shared library loads on a GPU-less box, exports every symbol
declared in include/scvae_hip.h, and the host-side graph builder (no kernels)
produces the reference's variable table."""
import ctypes
import os
import re

import pytest

from oracle import models as om

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scvae_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(scvae_[a-z0-9_]+)\s*\(", text)
    names = [n for n in names if n not in ("scvae_sync_fn",)]
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from scvae_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, "unbound symbol " + name
    assert sorted(_lib.SIGNATURES) == names
    assert lib.scvae_version() >= 1


def _table(lib, handle):
    from scvae_amd import _lib
    name = ctypes.create_string_buffer(_lib.NAME_MAX)
    off, rows, cols = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    out = []
    for i in range(lib.scvae_plan_param_count(handle)):
        assert lib.scvae_plan_param_info(
            handle, i, name, ctypes.byref(off), ctypes.byref(rows),
            ctypes.byref(cols)) == 0
        shape = (rows.value, cols.value) if cols.value else (rows.value,)
        out.append((name.value.decode(), off.value, shape))
    return out


@pytest.mark.parametrize("model_type,likelihood,bn", [
    ("VAE", "negative binomial", True), ("VAE", "poisson", False),
    ("GMVAE", "zero-inflated negative binomial", True)])
def test_plan_layout_matches_reference_variable_order(model_type, likelihood,
                                                      bn):
    from scvae_amd import _lib
    lib = _lib.load()
    cfg = _lib.ModelConfig()
    cfg.model_type = _lib.MODEL_GMVAE if model_type == "GMVAE" else 0
    cfg.feature_size, cfg.latent_size, cfg.n_hidden = 321, 7, 2
    cfg.hidden[0], cfg.hidden[1] = 40, 30
    cfg.likelihood = _lib.LIKELIHOOD_KINDS[likelihood][0]
    cfg.batch_norm = int(bn)
    cfg.n_clusters = 5
    cfg.kl_weight = 1.0
    handle = ctypes.c_void_p()
    assert lib.scvae_plan_create(ctypes.byref(cfg), ctypes.byref(handle)) == 0
    try:
        table = _table(lib, handle)
        ocfg = om.ModelConfig(feature_size=321, latent_size=7,
                              hidden_sizes=(40, 30), likelihood=likelihood,
                              minibatch_normalisation=bn, n_clusters=5)
        shapes = (om.gmvae_parameter_shapes(ocfg) if model_type == "GMVAE"
                  else om.vae_parameter_shapes(ocfg))
        assert [(n, s) for n, _, s in table] == [
            (n, tuple(s)) for n, s in shapes.items()]
        # 256-byte aligned, non-overlapping
        end = 0
        for _, off, shape in table:
            assert off % 64 == 0 and off >= end
            size = 1
            for s in shape:
                size *= s
            end = off + size
        assert lib.scvae_plan_param_floats(handle) >= end
        nbytes = lib.scvae_plan_workspace_bytes(handle, 100, 2)
        assert nbytes > 0
        assert lib.scvae_plan_workspace_bytes(handle, 200, 2) > nbytes
    finally:
        lib.scvae_plan_destroy(handle)


def test_bad_arguments_are_reported_not_crashed():
    from scvae_amd import _lib
    lib = _lib.load()
    cfg = _lib.ModelConfig()
    cfg.feature_size = 0
    handle = ctypes.c_void_p()
    assert lib.scvae_plan_create(ctypes.byref(cfg), ctypes.byref(handle)) == -1
    assert b"bad argument" in lib.scvae_last_error()
    assert lib.scvae_gemm(0, 0, None, None, None, None, 4, 4, 4, 4, 4, 4, 0, 0,
                          None, 0, None) == -1


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scvae_amd import _lib
    from scvae_amd.engine import Engine
    with pytest.raises(_lib.HipLibraryError, match="no CPU path"):
        Engine(10, 2, [4], "poisson")


def _c_layout(struct, fields, tmp_path):
    """sizeof / offsetof of ``struct`` as the C compiler lays it out from the header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    source = tmp_path / "layout.c"
    lines = ['#include <stddef.h>', '#include <stdio.h>',
             '#include "scvae_hip.h"', 'int main(void) {',
             '  printf("%zu\\n", sizeof({}));'.format(struct)]
    for name in fields:
        lines.append('  printf("%zu\\n", offsetof({}, {}));'.format(struct, name))
    lines += ['  return 0;', '}']
    source.write_text("\n".join(lines))
    binary = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"),
                    str(source), "-o", str(binary)], check=True)
    out = subprocess.run([str(binary)], check=True, capture_output=True,
                         text=True).stdout.split()
    return int(out[0]), [int(v) for v in out[1:]]


@pytest.mark.parametrize("struct,ctype", [("scvae_step_args", "StepArgs"),
                                          ("scvae_side_work", "SideWork"),
                                          ("scvae_model_config", "ModelConfig")])
def test_ctypes_structures_have_the_layout_of_the_header(tmp_path, struct, ctype):
    """The ctypes mirrors of the C structs (``scvae_amd/_lib.py``) against the
    layout a C compiler gives ``include/scvae_hip.h``: same size, same field
    order, same offsets -- the binding a maintainer of the reference would add
    passes these structs by pointer."""
    from scvae_amd import _lib
    mirror = getattr(_lib, ctype)
    names = [f[0] for f in mirror._fields_]
    size, offsets = _c_layout(struct, names, tmp_path)
    assert ctypes.sizeof(mirror) == size
    assert [getattr(mirror, n).offset for n in names] == offsets
