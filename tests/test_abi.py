"""not-gpu: the C-ABI library loads on a GPU-less box and exports every symbol include/gpe_hip.h declares; host-only
entry points behave; compute entry points are NOT called here."""
import ctypes
import os

import pytest

import gpe_amd
from gpe_amd import _lib


def test_header_declares_expected_surface():
    sigs = _lib.parse_header()
    assert len(sigs) >= 30
    for must in ('gpe_knn', 'gpe_linear', 'gpe_redgemm', 'gpe_edge_mlp_fwd', 'gpe_edge_mlp_bwd', 'gpe_edge_redgemm',
                 'gpe_edge_gather_stats', 'gpe_bn_finalize', 'gpe_lstm_step_fwd', 'gpe_lstm_cell_bwd',
                 'gpe_segment_mean_fwd', 'gpe_pack_weight'):
        assert must in sigs
    # every compute entry point ends with the stream argument (a pointer) and returns int
    for name, (res, args) in sigs.items():
        if name in ('gpe_abi_version', 'gpe_packed_size', 'gpe_packed_gates_size', 'gpe_redgemm_ws',
                    'gpe_stats_blocks', 'gpe_point_sums_blocks', 'gpe_debug_set', 'gpe_math_set', 'gpe_math_get',
                    'gpe_attn_pool_ws', 'gpe_packed_ngates_size', 'gpe_rnn_seq_bwd_ws', 'gpe_rnn_seq_fwd_ws', 'gpe_debug_get', 'gpe_reserve_cus_set', 'gpe_f16x3_min_rows',
                    'gpe_f16x3_min_rows_set', 'gpe_edge_ws_bytes', 'gpe_knn_ws_bytes', 'gpe_packed_planes_size', 'gpe_edge_lazy_dz3_ok',
                    'gpe_pack_job_blocks', 'gpe_adam_hyper'):
            continue
        assert res == 'i' and args[-1] == 'p', name


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(raw, name), 'libgpe_hip.so does not export %s' % name
    _lib.lib()      # binds argtypes for all of them


def test_math_mode_switch_is_host_only():
    assert _lib.get_math() in _lib.MATH_MODES
    prev = _lib.set_math('bf16x3')
    assert _lib.get_math() == 'bf16x3' and _lib.lib().gpe_math_set(7) == -22
    assert _lib.set_math('mixed') == 'bf16x3' and _lib.set_math('bf16x3') == 'mixed'
    assert _lib.set_math(prev) == 'bf16x3' and _lib.get_math() == prev


def test_host_only_queries():
    l = _lib.lib()
    assert l.gpe_abi_version() == 7
    # caller-owned workspaces (ABI version 4: the library allocates nothing): sizes are host-only queries
    fixed = l.gpe_edge_ws_bytes(32, 2048, 16, 400)
    assert fixed > 64 * 512 * 4 and l.gpe_edge_ws_bytes(1, 10, 5, 4) == fixed              # k <= 16: no pseudo-point rows
    assert l.gpe_edge_ws_bytes(32, 4096, 20, 400) > fixed + 32 * 4096 * 5 * 400 * 4          # k = 20: 5 pseudo-points per point
    assert l.gpe_edge_ws_bytes(0, 1, 1, 1) == -22
    assert l.gpe_knn_ws_bytes(32, 2048, 150, 16) >= 32 * 2048 * (32 * 8 + 4)
    # the f16x3 size gate is part of the arithmetic mode: settable, restorable
    gate = l.gpe_f16x3_min_rows()
    assert gate == 32768 and l.gpe_f16x3_min_rows_set(0) == gate and l.gpe_f16x3_min_rows() == 0
    assert l.gpe_f16x3_min_rows_set(gate) == 0 and l.gpe_f16x3_min_rows_set(-1) == -22
    assert l.gpe_packed_size(200, 200) == 208 * 208
    assert l.gpe_packed_size(7, 3) == 16 * 16
    assert l.gpe_packed_gates_size(250, 250) == 64 * 16 * 256
    assert l.gpe_stats_blocks() > 0 and l.gpe_point_sums_blocks() > 0
    assert l.gpe_redgemm_ws(200, 200) > 200 * 200
    assert l.gpe_redgemm_ws(1000, 250) > 1000 * 250


def test_cu_reservation_is_host_only():
    """gpe_reserve_cus_set (round 6): compute units left out of every persistent launch for a concurrent collective; host-only state."""
    l = _lib.lib()
    assert l.gpe_reserve_cus_set(16) == 0 and l.gpe_reserve_cus_set(0) == 16
    assert l.gpe_reserve_cus_set(-1) == -22 and l.gpe_reserve_cus_set(500) == -22
    import gpe_amd
    assert gpe_amd.set_reserved_cus(8) == 0 and gpe_amd.set_reserved_cus(0) == 8


def test_bad_arguments_are_rejected_without_a_gpu():
    l = _lib.lib()
    # NULL pointers / bad dims must come back as -EINVAL before any launch is attempted
    assert l.gpe_knn(None, 1, 8, 3, 3, 4, None, None, None, None, None, 0, None) == -22
    assert l.gpe_linear(None, 0, 0, 0, None, None, None, 0, 0, 0, None, 0, 0, 0, 4, 4, 4, 0, None) == -22
    assert l.gpe_edge_gather_stats(None, 0, 0, None, 1, 1, 1, None, None) == -22
    assert l.gpe_edge_pq_amax(None, 0, 0, 0, None, None, 0, None) == -22
    assert l.gpe_absmax(None, 0, 0, 0, None, None) == -22


def test_product_path_has_no_cpu_fallback():
    import torch
    x = torch.randn(8, 3)
    with pytest.raises(RuntimeError, match='no CPU path'):
        gpe_amd.ops.knn(x, 1, 8, 4)
    w = torch.randn(4, 3)
    with pytest.raises(RuntimeError, match='no CPU path'):
        gpe_amd.ops.linear(x, w, None)
