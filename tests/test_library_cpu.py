"""CPU: the C-ABI library builds, loads, exports every symbol include/vpr_amd.h declares, and
rejects bad arguments before touching a device (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import vpr_amd
    vpr_amd.build_library()          # hipcc cross-compiles gfx950 without a GPU
    from vpr_amd import _lib
    return _lib.lib()


def test_header_and_binding_agree(lib):
    from vpr_amd import _lib
    header = open(os.path.join(ROOT, "include", "vpr_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vpr_[a-z0-9_]+)\s*\(", header))
    declared -= {"vpr_status", "vpr_salad_weights", "vpr_salad_weights_f32"}
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libvpr_amd.so"
    assert lib.vpr_abi_version() == _lib.ABI_VERSION


def test_status_strings(lib):
    assert lib.vpr_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4, 17):
        assert len(lib.vpr_status_string(code)) > 0


def test_workspace_queries(lib):
    assert lib.vpr_knn_workspace_bytes(64, 100000, 8448, 10) >= 64 * 100000 * 4
    assert lib.vpr_knn_workspace_bytes(64, 1000, 8447, 10) == 0        # D % 64 != 0
    assert lib.vpr_knn_workspace_bytes(64, 1000, 8448, 65) == 0        # k > 64
    assert lib.vpr_knn_workspace_bytes(0, 1000, 8448, 10) == 0
    assert lib.vpr_salad_workspace_bytes(64, 256, 1024, 64, 128, 256, 512) >= 64 * 256 * 1024 * 2
    assert lib.vpr_pose_head_workspace_bytes(64, 8448, 512, 4) > 0
    assert lib.vpr_pose_head_workspace_bytes(64, 768, 0, 2) > 0


def test_invalid_arguments_are_rejected_without_a_device(lib):
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_char * 4096)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.vpr_knn_topk(null, null, 1, 1, 64, 1, 0, null, null, null, 0, null) in (-1, -3)
    assert lib.vpr_knn_topk(p, p, 1, 10, 65, 1, 0, p, p, p, 4096, null) == -2           # D % 64
    assert lib.vpr_knn_topk(p, p, 1, 10, 64, 1, 0, p, p, p, 16, null) == -3             # workspace too small
    assert lib.vpr_topk_merge(null, null, 2, 1, 1, null, null, null) == -1
    assert lib.vpr_topk_merge(p, p, 5000, 1, 1, p, p, null) == -2
    assert lib.vpr_pose_head(null, null, null, null, null, null, 1, 64, 32, 2, -1, null, 0, null) == -1
    assert lib.vpr_pose_head(p, p, p, p, p, p, 1, 65, 32, 2, -1, p, 4096, null) == -2    # D % 16
    assert lib.vpr_pose_head(p, p, p, p, p, p, 1, 64, 32, 9, -1, p, 4096, null) == -2    # n_out > 8
    assert lib.vpr_ln_meanpool_head(p, 0, 1, 4, 100, p, p, 1e-5, p, null, null, 0, -1, null, null) == -2   # H unsupported
    assert lib.vpr_gemm_nt_bf16(p, 64, 0, 0, p, 64, null, 0, p, 8, 0, 8, 8, 60, null) == -2   # K % 64
    assert lib.vpr_salad_sinkhorn_aggregate(p, p, p, 1, 100, 64, 128, 256, 1.0, 3, p, null, null) == -2
    assert lib.vpr_f32_to_bf16(null, null, 4, null) == -1
    # head-only fine-tuning step: arguments are judged before anything is launched
    hp = (1e-5, 0.9, 0.999, 1e-8, 1e-2, 0, 0.0)           # lr, betas, eps, weight decay, VPR_LOSS_MSE, (delta unused)
    ws = lib.vpr_head_train_workspace_bytes(16, 64, 32, 2)
    assert ws > 0 and lib.vpr_head_train_workspace_bytes(65, 64, 32, 2) == 0 and lib.vpr_head_train_workspace_bytes(16, 60, 32, 2) == 0
    assert lib.vpr_head_train_workspace_bytes(16, 64, 48, 2) == 0 and lib.vpr_head_train_workspace_bytes(16, 64, 32, 9) == 0
    assert lib.vpr_head_train_state_floats(8448, 512, 2) == 512 * 8448 + 512 + 2 * 512 + 2
    assert lib.vpr_head_train_step(null, 64, null, null, 2, 16, 64, 32, 2, null, null, null, null, null, null, 1, *hp, null, null, 0, null) == -1
    assert lib.vpr_head_train_step(p, 64, null, p, 2, 16, 64, 32, 2, p, p, p, p, p, p, 0, *hp, null, p, 4096, null) == -1        # step < 1
    assert lib.vpr_head_train_step(p, 64, null, p, 2, 16, 64, 32, 2, p, p, p, p, p, p, 1, 1e-5, 1.0, 0.999, 1e-8, 0.0, 0, 0.0, null, p, 4096, null) == -1   # beta1 = 1
    assert lib.vpr_head_train_step(p, 64, null, p, 2, 16, 64, 32, 2, p, p, p, p, p, p, 1, 1e-5, 0.9, 0.999, 1e-8, 0.0, 1, 0.0, null, p, 4096, null) == -1    # Huber needs delta > 0
    assert lib.vpr_head_train_step(p, 64, null, p, 2, 16, 64, 32, 2, p, p, p, p, p, p, 1, 1e-5, 0.9, 0.999, 1e-8, 0.0, 2, 1.0, null, p, 4096, null) == -1    # unknown loss kind
    assert lib.vpr_head_train_step(p, 64, null, p, 2, 65, 64, 32, 2, p, p, p, p, p, p, 1, *hp, null, p, 4096, null) == -2        # B > 64
    assert lib.vpr_head_train_step(p, 62, null, p, 2, 16, 64, 32, 2, p, p, p, p, p, p, 1, *hp, null, p, 4096, null) == -1        # x_stride < D
    assert lib.vpr_head_train_step(p, 64, null, p, 2, 16, 64, 32, 2, p, p, p, p, p, p, 1, *hp, null, p, 16, null) == -3          # workspace too small
    assert lib.vpr_head_train_epoch(p, 64, null, 40, 16, p, 2, 64, 32, 2, p, p, p, p, p, p, 1, *hp, null, p, 4096, null) == -1    # no order
    assert lib.vpr_head_train_epoch(p, 64, p, 100, 65, p, 2, 64, 32, 2, p, p, p, p, p, p, 1, *hp, null, p, 1 << 20, null) == -2   # batch > 64
    assert lib.vpr_head_train_epoch(p, 64, p, 40, 16, p, 2, 64, 32, 2, p, p, p, p, p, p, 1, *hp, null, p, 16, null) == -3
    assert lib.vpr_f32_to_bf16(p, p, 0, null) == 0


def test_ops_refuse_cpu_tensors():
    import torch
    from vpr_amd import ops
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.knn_topk(torch.zeros(1, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16), 1)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.pose_head(torch.zeros(1, 64), None, None, torch.zeros(2, 64), torch.zeros(2))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vpr_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "library_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_lds_swizzle_is_conflict_free():
    """Host re-derivation of vpr_common.h::tile_off for every ds_read_b128 lane group of both
    MFMA operand maps (bank = (addr/4) % 64; a 16-lane group must hit 16 distinct 16-B slots)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    off = lambda row, chunk: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
    for kk in (0, 1):                                   # 16x16x32: row = lane&15, chunk = lane>>4 (+4)
        for g in groups:
            slots = {(off(l & 15, (l >> 4) + 4 * kk) // 16) % 16 for l in g}
            assert len(slots) == 16
    for s in range(4):                                  # 32x32x16: row = lane&31, chunk = lane>>5 (+2s)
        for g in groups:
            slots = {(off(l & 31, (l >> 5) + 2 * s) // 16) % 16 for l in g}
            assert len(slots) == 16
    # the staging side writes physical slot (row, p) with logical chunk p ^ swz(row): a bijection per row
    for row in range(16):
        assert sorted((p ^ ((row >> 1) & 7)) for p in range(8)) == list(range(8))


def test_tuning_switches_are_read_once_at_load(lib, monkeypatch):
    """include/vpr_amd.h, "State the library keeps": the VPR_* switches are read from the environment when the library
    is loaded; a later setenv() has no effect, vpr_tuning_set() is the only way to change one, unknown names are refused."""
    from vpr_amd import _lib
    before = _lib.tuning_get("VPR_KNN_VARIANT")
    monkeypatch.setenv("VPR_KNN_VARIANT", "6")
    monkeypatch.setenv("VPR_POSE_KS", "9")
    assert _lib.tuning_get("VPR_KNN_VARIANT") == before               # the environment is not consulted again
    assert lib.vpr_knn_scores_kernel_name(0, 64, 100000) == b"vpr::knn_scores_kernel<false, 208, 2, 4>" or before not in (None, 0)
    with _lib.tuning(VPR_KNN_VARIANT=1):
        assert _lib.tuning_get("VPR_KNN_VARIANT") == 1
        assert lib.vpr_knn_scores_kernel_name(0, 64, 100000) == b"vpr::knn_scores_kernel<false, 208, 2, 0>"
    assert _lib.tuning_get("VPR_KNN_VARIANT") == before
    v = ctypes.c_int(0)
    assert lib.vpr_tuning_get(b"VPR_NOT_A_SWITCH", ctypes.byref(v)) == -1
    assert lib.vpr_tuning_set(b"VPR_NOT_A_SWITCH", 1, 0) == -1
    assert lib.vpr_tuning_set(None, 1, 0) == -1


def test_tuning_switches_come_from_the_environment_of_the_loading_process():
    """A fresh process with VPR_KNN_VARIANT=1 in its environment loads a library that reports 1 (and picks that kernel)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import vpr_amd; from vpr_amd import _lib; "
            "print(_lib.tuning_get('VPR_KNN_VARIANT'), _lib.tuning_get('VPR_POSE_KS'), "
            "_lib.lib().vpr_knn_scores_kernel_name(0, 64, 100000).decode())" % ROOT)
    env = dict(os.environ, VPR_KNN_VARIANT="1")
    env.pop("VPR_POSE_KS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == "1" and out[1] == "None" and "208, 2, 0" in " ".join(out[2:])
