"""GPU parity tests, op by op: every HIP kernel (called through the C ABI) against the oracle on the
same seeded inputs.  Integer / index outputs must be bit-exact; fp32-mode floats within the
tolerance written at each check; bf16-mode is compared with the oracle run on bf16-rounded operands."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as G
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
O = G.O


@pytest.fixture(scope="module")
def drn():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    pkg = load_package()
    pkg._cabi.lib()  # raises if the HIP library is missing: no fallback
    import importlib

    return importlib.import_module("drn_wsod_pytorch_amd.ops")


DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def _rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)


def _q(x, dtype):
    """value an operand has once stored in the compute dtype (fp32 copy of it)"""
    return x.to(dtype).float()


def _padded(x2d, dtype, drn):
    """[R, K] fp32 cpu -> device tensor [R, kpad(K)] of dtype, zero padded"""
    r, k = x2d.shape
    out = torch.zeros((r, drn.kpad(k, dtype)), dtype=dtype, device=DEV)
    out[:, :k] = x2d.to(DEV).to(dtype)
    return out


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(200, 103, 192), (64, 64, 64), (333, 257, 1000), (2048, 128, 4096), (5, 7, 54)])
def test_gemm_nt(drn, dtype, shape):
    M, N, K = shape
    A, B = _rnd((M, K), 1), _rnd((N, K), 2)
    Ad, Bd = _padded(A, dtype, drn), _padded(B, dtype, drn)
    Kp = Ad.shape[1]
    ref = (_q(A, dtype).double() @ _q(B, dtype).double().t())
    mag = (_q(A, dtype).abs().double() @ _q(B, dtype).abs().double().t())
    for splits in (1, 3):
        Cd = drn.gemm_nt(Ad, Bd, M, N, Kp, splits=splits)
        torch.cuda.synchronize()
        got = Cd.sum(0).cpu().double()
        # fp32 accumulation of exact products: error <= ~K * 2^-24 * sum|a*b| (loose factor 4)
        tol = 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6
        assert ((got - ref).abs() <= tol).all(), float(((got - ref).abs() / (mag + 1e-9)).max())
    # accumulate flag: C += A B^T
    C0 = _rnd((M, N), 3).to(DEV)
    Cacc = C0.clone().unsqueeze(0)
    drn.gemm_nt(Ad, Bd, M, N, Kp, out=Cacc, accumulate=True)
    assert torch.allclose(Cacc[0].cpu().double(), C0.cpu().double() + ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(300, 517, 256), (256, 256, 64), (2000, 600, 1024), (513, 259, 192)])
def test_gemm_nt_tile256(drn, dtype, shape):
    """the 256x256 LDS-DMA kernel, pinned: ragged M/N edges (hardware bounds check), split-K, accumulate"""
    M, N, K = shape
    A, B = _rnd((M, K), 4), _rnd((N, K), 5)
    Ad, Bd = _padded(A, dtype, drn), _padded(B, dtype, drn)
    Kp = Ad.shape[1]
    ref = _q(A, dtype).double() @ _q(B, dtype).double().t()
    mag = _q(A, dtype).abs().double() @ _q(B, dtype).abs().double().t()
    tol = 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6
    prev = drn.gemm_set_tile(256)
    try:
        for splits in (1, 2):
            got = drn.gemm_nt(Ad, Bd, M, N, Kp, splits=splits).sum(0).cpu().double()
            assert ((got - ref).abs() <= tol).all(), float(((got - ref).abs() / (mag + 1e-9)).max())
        C0 = _rnd((M, N), 6).to(DEV)
        Cacc = C0.clone().unsqueeze(0)
        drn.gemm_nt(Ad, Bd, M, N, Kp, out=Cacc, accumulate=True)
        assert torch.allclose(Cacc[0].cpu().double(), C0.cpu().double() + ref, rtol=1e-4, atol=1e-3)
        # A = I against an asymmetric B: catches a transposed / mis-mapped C write of the swapped-operand epilogue
        n = 256
        Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251)
        Cd = drn.gemm_nt(torch.eye(n).to(DEV), Bm.to(DEV), n, n, n)
        assert torch.equal(Cd[0].cpu(), Bm.t().contiguous())
    finally:
        drn.gemm_set_tile(prev)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,splits", [(2000, 4500, 512, 2), (1030, 17000, 192, 1), (2304, 8192, 256, 1)])
def test_gemm_persistent_equals_one_tile_grid(drn, dtype, M, N, K, splits):
    """gemm_nt256p_kernel (one resident workgroup per CU walking the (tile, K-split) list) against the one-tile grid of
    the same 256x256 kernel: same K order per output element, so every variant must be BIT-identical - plain fp32
    output with split-K, accumulate, bf16 output; more work items than CUs (ragged last round) and ragged M / N edges."""
    A, B = _rnd((M, K), 14), _rnd((N, K), 15)
    Ad, Bd = _padded(A, dtype, drn), _padded(B, dtype, drn)
    Kp = Ad.shape[1]
    C0 = _rnd((M, N), 16).to(DEV)
    prev_tile = drn.gemm_set_tile(256)
    prev = drn.tune(drn.TUNE_GEMM_PERSISTENT, 0)
    try:
        res = []
        for persistent in (0, 1):
            drn.tune(drn.TUNE_GEMM_PERSISTENT, persistent)
            out = [drn.gemm_nt(Ad, Bd, M, N, Kp, splits=splits).clone()]
            acc = C0.clone().unsqueeze(0)
            drn.gemm_nt(Ad, Bd, M, N, Kp, out=acc, accumulate=True)
            out.append(acc)
            o16 = torch.zeros((1, M, N), dtype=torch.bfloat16, device=DEV)
            drn.gemm_nt(Ad, Bd, M, N, Kp, out=o16)
            out.append(o16)
            res.append(out)
        torch.cuda.synchronize()
        for a, b in zip(*res):
            assert torch.equal(a, b)
        ref = _q(A, dtype).double() @ _q(B, dtype).double().t()
        mag = _q(A, dtype).abs().double() @ _q(B, dtype).abs().double().t()
        got = res[1][0].sum(0).cpu().double()
        assert ((got - ref).abs() <= 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6).all()
    finally:
        drn.tune(drn.TUNE_GEMM_PERSISTENT, prev)
        drn.gemm_set_tile(prev_tile)


@pytest.mark.parametrize("shapes", [((4096, 2048, 2048, 1, True), (2000, 2048, 4096, 4, False)),
                                    ((300, 500, 128, 1, False), (700, 260, 192, 2, False)),
                                    ((2000, 2048, 4096, 4, False), (256, 256, 64, 1, True)),
                                    ((1030, 4600, 320, 1, True), (1030, 4600, 320, 3, False))])
def test_gemm_nt_pair_equals_two_calls(drn, shapes):
    """drn_gemm_nt_pair: two independent NT GEMMs in one persistent launch (the workgroups of every XCD divided between
    them by work) against the two separate 256x256 launches: same tiles, slab order and MFMA per output element, so
    BIT-identical - the fc7 weight-gradient / input-gradient pair of the bench shape, ragged shapes, very unequal work,
    accumulate and split-K outputs."""
    dtype = torch.bfloat16
    prev_tile = drn.gemm_set_tile(256)
    try:
        gs, refs = [], []
        for i, (M, N, K, splits, acc) in enumerate(shapes):
            A, B = _padded(_rnd((M, K), 41 + i), dtype, drn), _padded(_rnd((N, K), 51 + i), dtype, drn)
            Kp = A.shape[1]
            c0 = _rnd((splits, M, N), 61 + i).to(DEV)
            ref = c0.clone()
            drn.gemm_nt(A, B, M, N, Kp, out=ref, splits=splits, accumulate=acc)
            refs.append(ref)
            gs.append(dict(A=A, B=B, M=M, N=N, K=Kp, out=c0.clone(), splits=splits, accumulate=acc))
        drn.gemm_nt_pair(gs[0], gs[1])
        torch.cuda.synchronize()
        for g, ref in zip(gs, refs):
            assert torch.equal(g["out"], ref)
    finally:
        drn.gemm_set_tile(prev_tile)


@pytest.mark.parametrize("M,N,Kb,splits", [(256, 256, 64, 1), (1024, 6400, 2000, 1), (700, 1000, 130, 1), (2048, 1024, 2000, 1),
                                           (512, 33000, 200, 1), (1000, 2304, 700, 3), (2048, 25088, 37, 1)])
def test_gemm_tn_equals_nt_on_the_transpose(drn, M, N, Kb, splits):
    """drn_gemm_tn (second operand K-major, read through ds_read_b64_tr_b16) against drn_gemm_nt on a materialised
    transpose with the 256x256 ping-pong kernel: same tile, slab order and MFMA operands per output element, so the
    results are BIT-identical - fp32 with split-K, accumulate, bf16 output; persistent and one-tile grids, ragged M / N,
    contraction lengths that are not a multiple of the 64-row slab (the missing rows read as zeros), a column slice of
    a wider matrix as Bt (the fc6 dW's main columns)."""
    dtype = torch.bfloat16
    K = (Kb + 63) // 64 * 64
    A = _rnd((M, Kb), 21)
    Bt = _rnd((Kb, N + 40), 22)  # wider than N: the operand is a column slice, its tail columns must not leak in
    Ad = torch.zeros((M, K), dtype=dtype, device=DEV)
    Ad[:, :Kb] = A.to(dtype)
    Btd = Bt.to(dtype).to(DEV).contiguous()
    Bd = torch.zeros((N, K), dtype=dtype, device=DEV)  # the materialised transpose, K padded with zeros
    Bd[:, :Kb] = Btd[:, :N].t()
    C0 = _rnd((M, N), 23).to(DEV)
    prev_tile = drn.gemm_set_tile(256)
    prev_pp = drn.tune(drn.TUNE_GEMM_PINGPONG, 1)
    prev_ts = drn.tune(drn.TUNE_GEMM_TAIL_SPLIT, 0)  # (the NT side would peel columns into the small-tile kernel: same bits, other kernel)
    try:
        nt = drn.gemm_nt(Ad, Bd, M, N, K, splits=splits)
        tn = drn.gemm_tn(Ad, Btd[:, :N], M, N, K, Kb, splits=splits)
        torch.cuda.synchronize()
        assert torch.equal(nt, tn), float((nt - tn).abs().max())
        a1, a2 = C0.clone().unsqueeze(0), C0.clone().unsqueeze(0)
        drn.gemm_nt(Ad, Bd, M, N, K, out=a1, accumulate=True)
        drn.gemm_tn(Ad, Btd[:, :N], M, N, K, Kb, out=a2, accumulate=True)
        o1 = torch.zeros((1, M, N), dtype=torch.bfloat16, device=DEV)
        o2 = torch.zeros((1, M, N), dtype=torch.bfloat16, device=DEV)
        drn.gemm_nt(Ad, Bd, M, N, K, out=o1)
        drn.gemm_tn(Ad, Btd[:, :N], M, N, K, Kb, out=o2)
        torch.cuda.synchronize()
        assert torch.equal(a1, a2) and torch.equal(o1, o2)
        ref = _q(A, dtype).double() @ _q(Bt[:, :N], dtype).double()
        mag = _q(A, dtype).abs().double() @ _q(Bt[:, :N], dtype).abs().double()
        assert ((tn.sum(0).cpu().double() - ref).abs() <= 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6).all()
    finally:
        drn.tune(drn.TUNE_GEMM_TAIL_SPLIT, prev_ts)
        drn.tune(drn.TUNE_GEMM_PINGPONG, prev_pp)
        drn.gemm_set_tile(prev_tile)


@pytest.mark.parametrize("M,N,K,splits", [(2000, 2048, 3136, 4), (256, 256, 64, 1), (1030, 17000, 192, 1), (300, 70000, 128, 1),
                                          (2304, 8192, 320, 3), (2000, 4500, 512, 2)])
def test_gemm_pingpong_bit_identical(drn, M, N, K, splits):
    """Round 3: the ping-pong mainloop of the bf16 256x256 kernels (DRN_TUNE_GEMM_PINGPONG: the two wave rows half a phase
    apart, half-tile LDS-DMA with counted waits) against the lock-step pipeline it replaces, in the one-tile grid AND the
    persistent kernel: the same MFMA sees the same K slabs in the same order for every output element, so fp32 split-K
    partials, accumulate and bf16 outputs must be BIT-identical.  Shapes: one slab, odd slab counts, slab ranges that
    differ per K-split, ragged M / N edges, more work items than CUs; repeated launches (a schedule race would show as
    run-to-run differences)."""
    dtype = torch.bfloat16
    A, B = _rnd((M, K), 44), _rnd((N, K), 45)
    Ad, Bd = _padded(A, dtype, drn), _padded(B, dtype, drn)
    Kp = Ad.shape[1]
    C0 = _rnd((M, N), 46).to(DEV)
    prev_tile = drn.gemm_set_tile(256)
    prev_p = drn.tune(drn.TUNE_GEMM_PERSISTENT, 1)
    prev = drn.tune(drn.TUNE_GEMM_PINGPONG, 1)
    try:
        res = {}
        for persistent in (0, 1):
            drn.tune(drn.TUNE_GEMM_PERSISTENT, persistent)
            for pp in (0, 1, 1, 1):
                drn.tune(drn.TUNE_GEMM_PINGPONG, pp)
                out = [drn.gemm_nt(Ad, Bd, M, N, Kp, splits=splits).clone()]
                acc = C0.clone().unsqueeze(0)
                drn.gemm_nt(Ad, Bd, M, N, Kp, out=acc, accumulate=True)
                out.append(acc)
                o16 = torch.zeros((1, M, N), dtype=torch.bfloat16, device=DEV)
                drn.gemm_nt(Ad, Bd, M, N, Kp, out=o16)
                out.append(o16)
                res.setdefault((persistent, pp), []).append(out)
        torch.cuda.synchronize()
        base = res[(0, 0)][0]
        for key, runs in res.items():
            for out in runs:
                for a, b in zip(base, out):
                    assert torch.equal(a, b), key
        ref = _q(A, dtype).double() @ _q(B, dtype).double().t()
        mag = _q(A, dtype).abs().double() @ _q(B, dtype).abs().double().t()
        got = res[(1, 1)][0][0].sum(0).cpu().double()
        assert ((got - ref).abs() <= 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6).all()
    finally:
        drn.tune(drn.TUNE_GEMM_PINGPONG, prev)
        drn.tune(drn.TUNE_GEMM_PERSISTENT, prev_p)
        drn.gemm_set_tile(prev_tile)


def test_gemm_joint_peel_of_row_slabs_bit_identical(drn):
    """run_fc1_tail's joint peel: two equal row slabs of one product [2048 x (68 * 256)] - each 4 x 68 = 256 + 16 tiles,
    so drn_gemm_nt peels the same 4 trailing tile columns off both - computed (a) slab by slab, each call peeling its
    own columns, (b) the peeled columns of ALL rows in one launch + the slabs' main columns (drn_gemm_nt_main_cols) as
    exact rounds.  Every output element sees the same K slabs in the same order: the bf16 buckets must be bit-equal."""
    M, N, K = 2048, 68 * 256, 256
    A, B = _rnd((M, K), 34), _rnd((N, K), 35)
    Ad, Bd = _padded(A, torch.bfloat16, drn), _padded(B, torch.bfloat16, drn)
    Kp = Ad.shape[1]
    prev_tile = drn.gemm_set_tile(0)
    try:
        n0 = drn.gemm_nt_main_cols(M // 2, N)
        assert n0 == 64 * 256 and drn.gemm_nt_main_cols(M // 2, n0) == n0  # the main columns are not peeled again
        a = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        b = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        for r0 in (0, M // 2):
            drn.gemm_nt(Ad[r0:r0 + M // 2], Bd, M // 2, N, Kp, out=a[r0:r0 + M // 2].unsqueeze(0))
        drn.gemm_nt(Ad, Bd[n0:], M, N - n0, Kp, out=b[:, n0:].unsqueeze(0))
        for r0 in (0, M // 2):
            drn.gemm_nt(Ad[r0:r0 + M // 2], Bd[:n0], M // 2, n0, Kp, out=b[r0:r0 + M // 2, :n0].unsqueeze(0))
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        ref = _q(A, torch.bfloat16).double() @ _q(B, torch.bfloat16).double().t()
        assert (b.cpu().double() - ref).abs().max() <= 2.0 ** -8 * ref.abs().max() + 1e-6
    finally:
        drn.gemm_set_tile(prev_tile)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1024, 68 * 256 - 50, 256), (2000, 36 * 256 - 3, 192)])
def test_gemm_tail_split_bit_identical(drn, dtype, M, N, K):
    """Tail balancing of the persistent 256x256 GEMM (DRN_TUNE_GEMM_TAIL_SPLIT): the tile columns of a nearly empty last
    round are peeled off and run on the small-tile kernel.  Both kernels consume K in the same 128-byte slabs with the
    same MFMA per output element, so fp32, accumulate and bf16 outputs must be BIT-identical with the split on and off
    (shapes chosen so that the split triggers: 4 x 68 = 256 + 16 tiles, 8 x 36 = 256 + 32 tiles, ragged N edge)."""
    A, B = _rnd((M, K), 24), _rnd((N, K), 25)
    Ad, Bd = _padded(A, dtype, drn), _padded(B, dtype, drn)
    Kp = Ad.shape[1]
    C0 = _rnd((M, N), 26).to(DEV)
    prev_tile = drn.gemm_set_tile(256)
    prev_p = drn.tune(drn.TUNE_GEMM_PERSISTENT, 1)
    prev = drn.tune(drn.TUNE_GEMM_TAIL_SPLIT, 0)
    try:
        res = []
        for split in (0, 1):
            drn.tune(drn.TUNE_GEMM_TAIL_SPLIT, split)
            out = [drn.gemm_nt(Ad, Bd, M, N, Kp).clone()]
            acc = C0.clone().unsqueeze(0)
            drn.gemm_nt(Ad, Bd, M, N, Kp, out=acc, accumulate=True)
            out.append(acc)
            o16 = torch.zeros((1, M, N), dtype=torch.bfloat16, device=DEV)
            drn.gemm_nt(Ad, Bd, M, N, Kp, out=o16)
            out.append(o16)
            res.append(out)
        torch.cuda.synchronize()
        for a, b in zip(*res):
            assert torch.equal(a, b)
        ref = _q(A, dtype).double() @ _q(B, dtype).double().t()
        mag = _q(A, dtype).abs().double() @ _q(B, dtype).abs().double().t()
        got = res[1][0].sum(0).cpu().double()
        assert ((got - ref).abs() <= 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6).all()
    finally:
        drn.tune(drn.TUNE_GEMM_TAIL_SPLIT, prev)
        drn.tune(drn.TUNE_GEMM_PERSISTENT, prev_p)
        drn.gemm_set_tile(prev_tile)


def test_stage_heads_inputs(drn):
    """drn_stage_heads_inputs: props <- rois[:, 1:5] and the label block copy, one launch (bit-exact copies; edge sizes)"""
    for M, n in ((2000, 41), (1, 3), (37, 0), (0, 5)):
        rois = torch.from_numpy(np.random.RandomState(M + 1).standard_normal((M, 5)).astype(np.float32)).to(DEV)
        props = torch.full((M, 4), -7.0, device=DEV)
        src = torch.arange(n, dtype=torch.int32, device=DEV) * 3 - 5
        dst = torch.full((n,), 99, dtype=torch.int32, device=DEV)
        drn.stage_heads_inputs(rois, props, src if n else None, dst if n else None)
        torch.cuda.synchronize()
        assert torch.equal(props, rois[:, 1:].contiguous())
        assert torch.equal(dst, src)


def test_gemm_asymmetric_identity(drn):
    """A = I with an ASYMMETRIC B catches a transposed C write (cdna guide rule 16)."""
    n = 128
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251  # exact in bf16? no -> use f32
    A = torch.eye(n)
    Cd = drn.gemm_nt(A.to(DEV), B.to(DEV), n, n, n)
    assert torch.equal(Cd[0].cpu(), B.t().contiguous())  # C[i][j] = sum_k I[i][k] B[j][k] = B[j][i]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_full_size_sampled(drn, dtype):
    """BASELINE configs[1] fc6 shape (R=2000 padded to 2048, K=50176, N=2048): spot-check 256 entries
    against fp64, plus linearity C(2A) = 2 C(A) (exact in binary floating point)."""
    M, N, K = 2000, 2048, 50176
    g = torch.Generator(device=DEV).manual_seed(0)
    A = (torch.randn((M, K), device=DEV, generator=g) * 0.5).to(dtype)
    B = (torch.randn((N, K), device=DEV, generator=g) * 0.02).to(dtype)
    C1 = drn.gemm_nt(A, B, M, N, K, splits=4).sum(0)
    C2 = drn.gemm_nt((A.float() * 2).to(dtype), B, M, N, K, splits=4).sum(0)
    assert torch.equal(C2, 2 * C1)
    rs = np.random.RandomState(0)
    ii, jj = rs.randint(0, M, 256), rs.randint(0, N, 256)
    a = A[torch.from_numpy(ii).to(DEV)].double()
    b = B[torch.from_numpy(jj).to(DEV)].double()
    ref = (a * b).sum(1)
    mag = (a.abs() * b.abs()).sum(1)
    got = C1[torch.from_numpy(ii).to(DEV), torch.from_numpy(jj).to(DEV)].double()
    assert ((got - ref).abs() <= 4 * 2.0 ** -24 * math.sqrt(K) * mag + 1e-6).all()


# ------------------------------------------------------------------------------------------- conv
def _pack_w(w, dtype, drn, cin_pad=None):
    cout, cin, kh, kw = w.shape
    cp = cin_pad or cin
    wp = torch.zeros((cout, kh, kw, cp))
    wp[..., :cin] = w.permute(0, 2, 3, 1)
    return _padded(wp.reshape(cout, -1), dtype, drn)


CONV_CASES = [
    # (N, H, W, Cin, Cout, k, stride, pad, dil, residual, relu)
    (1, 27, 27, 32, 48, 1, 1, 0, 1, False, True),
    (2, 14, 14, 64, 64, 3, 1, 1, 1, True, True),
    (1, 27, 27, 16, 40, 3, 1, 2, 2, False, True),
    (1, 33, 31, 3, 8, 3, 2, 1, 1, False, True),
    (1, 56, 56, 64, 256, 1, 1, 0, 1, True, False),
    (3, 9, 11, 8, 136, 3, 1, 1, 1, False, False),
    (1, 131, 160, 16, 64, 3, 1, 1, 1, False, True),   # large M, narrow output: the 128x64 tile
    (1, 131, 160, 16, 48, 1, 1, 0, 1, True, True),    # same with a ragged Cout and a residual
    (1, 150, 150, 32, 192, 3, 1, 1, 1, False, True),  # large M, 128x128 tile with a ragged second column tile
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_nhwc(drn, dtype, case):
    n, h, w, cin, cout, k, stride, pad, dil, has_res, relu = case
    x = _rnd((n, cin, h, w), 5)
    wt = _rnd((cout, cin, k, k), 6, math.sqrt(2.0 / (cin * k * k)))
    scale, bias = 0.8 + 0.2 * torch.rand(cout), _rnd((cout,), 7, 0.1)
    cin_pad = (8 if dtype == torch.bfloat16 else 4) if cin == 3 else cin
    xd = torch.zeros((n, h, w, cin_pad), dtype=dtype, device=DEV)
    xd[..., :cin] = x.permute(0, 2, 3, 1).to(DEV).to(dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, stride, pad, dil) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = None
    if has_res:
        res = _rnd(tuple(ref.shape), 8)
        ref = ref + _q(res, dtype)
        res = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    if relu:
        ref = F.relu(ref)
    y = drn.conv2d_nhwc(xd, _pack_w(wt, dtype, drn, cin_pad), cout, k, k, stride, pad, dil, scale.to(DEV), bias.to(DEV),
                        res, relu)
    got = y.float().cpu().permute(0, 3, 1, 2)
    if dtype == torch.float32:
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4), float((got - ref).abs().max())
    else:  # output rounded to bf16: half-ulp = 2^-9 relative
        assert torch.allclose(got, ref, rtol=2 ** -7, atol=2e-2), float((got - ref).abs().max())


@pytest.mark.parametrize("case", [(1, 203, 181, False, True),    # ragged tile edges in both directions (203 = 25*8+3, 181 = 5*32+21)
                                  (2, 184, 192, True, True),     # two images, residual
                                  (1, 181, 203, True, False)])
def test_conv3x3_c64_patch_kernel(drn, case):
    """conv3x3_c64_kernel (input patch + all weights resident in LDS; 3x3 / stride 1 / 64 -> 64 channels on maps of >= 32k
    pixels, bf16) against F.conv2d and - same k order, same MFMA - BIT-identical to the tiled kernel it replaces
    (DRN_TUNE_CONV_PATCH = 0); zero padding at every image border, ragged 8 x 32 pixel blocks."""
    n, h, w, has_res, relu = case
    dtype = torch.bfloat16
    x = _rnd((n, 64, h, w), 25)
    wt = _rnd((64, 64, 3, 3), 26, math.sqrt(2.0 / (64 * 9)))
    scale, bias = (0.8 + 0.2 * torch.rand(64)).to(DEV), _rnd((64,), 27, 0.1).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, 1, 1, 1) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1)
    res = None
    if has_res:
        res = _rnd(tuple(ref.shape), 28)
        ref = ref + _q(res, dtype)
        res = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    if relu:
        ref = F.relu(ref)
    wp = _pack_w(wt, dtype, drn, 64)
    prev = drn.tune(drn.TUNE_CONV_PATCH, 1)
    try:
        y1 = drn.conv2d_nhwc(xd, wp, 64, 3, 3, 1, 1, 1, scale, bias, res, relu)
        drn.tune(drn.TUNE_CONV_PATCH, 0)
        y0 = drn.conv2d_nhwc(xd, wp, 64, 3, 3, 1, 1, 1, scale, bias, res, relu)
    finally:
        drn.tune(drn.TUNE_CONV_PATCH, prev if prev else 1)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0)
    got = y1.float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=2e-2), float((got - ref).abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(1, 14, 14, 256, 256, 3, 1, 1, 1, True, True),     # res4 3x3 on 196 pixels (36 slabs)
                                  (1, 14, 14, 1024, 256, 1, 1, 0, 1, False, True),   # res4 conv1 (16 slabs)
                                  (1, 28, 28, 128, 128, 3, 1, 1, 1, False, False),   # res3 3x3
                                  (2, 13, 17, 72, 40, 3, 1, 2, 2, True, True),       # dilation, ragged Cout, two images
                                  (1, 21, 19, 64, 24, 3, 2, 1, 1, True, False),      # stride 2, Cout < 32
                                  (1, 9, 9, 512, 100, 1, 1, 0, 1, True, True)])      # Cout % 8 != 0: scalar epilogue
def test_conv2d_wave_k_split(drn, dtype, case):
    """conv_nhwc_ks_kernel (32x32 tile, the slab's four k-steps on four waves, fixed-order LDS reduction) serves the
    latency-bound small-map layers: against F.conv2d and against the 64x64 kernel; run-to-run identical"""
    n, h, w, cin, cout, k, stride, pad, dil, has_res, relu = case
    x = _rnd((n, cin, h, w), 15)
    wt = _rnd((cout, cin, k, k), 16, math.sqrt(2.0 / (cin * k * k)))
    scale, bias = (0.8 + 0.2 * torch.rand(cout)).to(DEV), _rnd((cout,), 17, 0.1).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, stride, pad, dil) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1)
    res = None
    if has_res:
        res = _rnd(tuple(ref.shape), 18)
        ref = ref + _q(res, dtype)
        res = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    if relu:
        ref = F.relu(ref)
    wp = _pack_w(wt, dtype, drn, cin)
    run = lambda: drn.conv2d_nhwc(xd, wp, cout, k, k, stride, pad, dil, scale, bias, res, relu)
    assert drn.tune(drn.TUNE_CONV_KSPLIT, 0) == 1
    try:
        tiled = run()
    finally:
        drn.tune(drn.TUNE_CONV_KSPLIT, 1)
    ys = [run() for _ in range(3)]
    for y in ys[1:]:
        assert torch.equal(y, ys[0])
    got = ys[0].float().cpu().permute(0, 3, 1, 2)
    if dtype == torch.float32:
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4), float((got - ref).abs().max())
        assert torch.allclose(ys[0], tiled, rtol=1e-5, atol=1e-5)
        assert not torch.equal(ys[0], tiled) or cin * k * k <= 128  # another kernel, another summation order
    else:
        assert torch.allclose(got, ref, rtol=2 ** -7, atol=2e-2), float((got - ref).abs().max())
        assert (ys[0].float() - tiled.float()).abs().max() <= 2 ** -7 * float(tiled.float().abs().max())


@pytest.mark.parametrize("case", [(1, 50, 76, 256, 256, 3, 1, 1, 1, False, True),     # res4 3x3 at 800x1216 (36 slabs, 240 64x64 tiles)
                                  (1, 50, 76, 512, 512, 3, 1, 2, 2, False, True),     # dilated res5 3x3 of the DC5 trunk (72 slabs)
                                  (1, 50, 76, 1024, 256, 1, 1, 0, 1, False, True),    # res4 conv1
                                  (1, 50, 76, 256, 1024, 1, 1, 0, 1, True, True),     # res4 conv3 + shortcut (4 slabs)
                                  (1, 100, 152, 128, 128, 3, 1, 1, 1, False, True),   # res3 3x3
                                  (1, 100, 152, 128, 512, 1, 1, 0, 1, True, True),    # res3 conv3: 2 slabs (< ring depth)
                                  (1, 131, 97, 64, 72, 1, 1, 0, 1, True, False),      # ONE slab, ragged M and a ragged column tile
                                  (2, 61, 67, 64, 136, 3, 2, 1, 1, True, True),       # stride 2, two images, Cout % 64 != 0
                                  (3, 45, 52, 192, 200, 3, 1, 3, 3, False, False),    # dilation 3, three slabs per tap
                                  (4, 28, 28, 512, 512, 3, 1, 2, 2, False, True),     # a trunk group of four 224x224 images: dilated res5
                                  (4, 28, 28, 512, 2048, 1, 1, 0, 1, True, True)])    # ... and its 1x1 to 2048 channels
def test_conv_ring_kernels(drn, case):
    """conv_ring_kernel<64x64 | 128x128> (register ring of prefetched K slabs + two LDS stages behind ONE LDS-only barrier per
    slab, im2col by per-lane buffer offsets on a per-row tap-validity mask; the bf16 trunk layers beyond the small maps) == the
    register-staged tiled kernel bit for bit (same slab order, k-steps and MFMA per output element) for every tile shape, and
    within bf16 rounding of F.conv2d; zero padding at every border, ragged M / Cout tiles, fewer slabs than ring slots,
    stride, dilation, residual."""
    n, h, w, cin, cout, k, stride, pad, dil, has_res, relu = case
    dtype = torch.bfloat16
    x = _rnd((n, cin, h, w), 35)
    wt = _rnd((cout, cin, k, k), 36, math.sqrt(2.0 / (cin * k * k)))
    scale, bias = (0.8 + 0.2 * torch.rand(cout)).to(DEV), _rnd((cout,), 37, 0.1).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, stride, pad, dil) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1)
    res = None
    if has_res:
        res = _rnd(tuple(ref.shape), 38)
        ref = ref + _q(res, dtype)
        res = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    if relu:
        ref = F.relu(ref)
    wp = _pack_w(wt, dtype, drn, cin)
    run = lambda: drn.conv2d_nhwc(xd, wp, cout, k, k, stride, pad, dil, scale, bias, res, relu)
    assert drn.tune(drn.TUNE_CONV_RING, 0) == 1
    k2 = drn.tune(drn.TUNE_CONV_K2_TILES, 0)
    ks = drn.tune(drn.TUNE_CONV_KSPLIT, 0)
    p8 = drn.tune(drn.TUNE_PP8, 0)
    try:
        tiled = run()  # conv_nhwc_kernel<64x64 | 128x128>
        ys = {}
        for pin in (64, 128, 1):
            drn.tune(drn.TUNE_CONV_RING, pin)
            ys[pin] = run()
        drn.tune(drn.TUNE_CONV_RING, 64)
        again = run()
    finally:
        drn.tune(drn.TUNE_CONV_RING, 1)
        drn.tune(drn.TUNE_CONV_K2_TILES, k2)
        drn.tune(drn.TUNE_CONV_KSPLIT, ks)
        drn.tune(drn.TUNE_PP8, p8)
    torch.cuda.synchronize()
    for pin, y in ys.items():
        assert torch.equal(y, tiled), (pin, float((y.float() - tiled.float()).abs().max()))
    assert torch.equal(again, tiled)
    got = tiled.float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=2e-2), float((got - ref).abs().max())


PP8_CASES = [(1, 99, 151, 256, 256, 3, 1, 2, 2, False, True),    # dilated-C5 res4 3x3 at 800x1216: 117 x 2 tiles, 36 slabs
             (1, 50, 76, 512, 512, 3, 1, 2, 2, False, True),     # dilated res5 3x3 (72 slabs)
             (1, 50, 76, 1024, 256, 1, 1, 0, 1, False, True),    # res4 conv1 (16 slabs)
             (1, 50, 76, 256, 1024, 1, 1, 0, 1, True, True),     # res4 conv3 + shortcut (4 slabs)
             (1, 100, 152, 128, 128, 3, 1, 1, 1, False, True),   # res3 3x3
             (1, 100, 152, 128, 512, 1, 1, 0, 1, True, True),    # two slabs: fewer than the ring holds
             (1, 131, 97, 64, 72, 1, 1, 0, 1, True, False),      # ONE slab, ragged M and a ragged column tile
             (2, 61, 67, 64, 136, 3, 2, 1, 1, True, True),       # stride 2, two images, Cout % 128 != 0
             (3, 45, 52, 192, 200, 3, 1, 3, 3, False, False),    # dilation 3, three slabs per tap
             (4, 28, 28, 512, 2048, 1, 1, 0, 1, True, True)]     # a trunk group of four 224x224 images


@pytest.mark.parametrize("case", PP8_CASES)
def test_pp8_conv_kernel(drn, case):
    """pp8_kernel<conv> (pp8.hip: 128x128 or 256x128 tile on EIGHT waves, the two waves of a SIMD half a phase apart, LDS-DMA ring
    of 3 / 4 / 5 stages, im2col by per-lane source offsets with out-of-range zero fill) == the register-staged tiled kernel bit
    for bit (same slab order, k-steps and MFMA per output element) at every ring depth, schedule variant and tile shape, run to
    run, and within bf16 rounding of F.conv2d; zero padding at every border, ragged M / Cout tiles, fewer slabs than ring stages, stride, dilation, residual."""
    n, h, w, cin, cout, k, stride, pad, dil, has_res, relu = case
    dtype = torch.bfloat16
    x = _rnd((n, cin, h, w), 75)
    wt = _rnd((cout, cin, k, k), 76, math.sqrt(2.0 / (cin * k * k)))
    scale, bias = (0.8 + 0.2 * torch.rand(cout)).to(DEV), _rnd((cout,), 77, 0.1).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, stride, pad, dil) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1)
    res = None
    if has_res:
        res = _rnd(tuple(ref.shape), 78)
        ref = ref + _q(res, dtype)
        res = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    if relu:
        ref = F.relu(ref)
    wp = _pack_w(wt, dtype, drn, cin)
    run = lambda: drn.conv2d_nhwc(xd, wp, cout, k, k, stride, pad, dil, scale, bias, res, relu)
    assert drn.tune(drn.TUNE_PP8, 0) == 1
    ring = drn.tune(drn.TUNE_CONV_RING, 0)
    k2 = drn.tune(drn.TUNE_CONV_K2_TILES, 0)
    ks = drn.tune(drn.TUNE_CONV_KSPLIT, 0)
    cpp = drn.tune(drn.TUNE_CONV_PP, 0)
    try:
        tiled = run()  # conv_nhwc_kernel<64x64 | 128x128>
        drn.tune(drn.TUNE_PP8, 2)
        ys = {}
        for stages, variant, wide in ((3, 0, 0), (4, 0, 0), (4, 1, 0), (5, 0, 0), (5, 2, 0), (5, 5, 0), (5, 1, 0), (5, 1, 2), (5, 0, 2),
                                      (5, 1, 1)):
            drn.tune(drn.TUNE_PP8_STAGES, stages)
            drn.tune(drn.TUNE_PP8_VARIANT, variant)
            drn.tune(drn.TUNE_PP8_WIDE, wide)  # 2: the 256x128 form (three 48-KB stages) whatever the tile count
            drn.tune(drn.TUNE_PP8_WIDE_VARIANT, 5 if variant else 0)  # (its two DMA placements)
            ys[(stages, variant, wide)] = [run() for _ in range(3)]
    finally:
        drn.tune(drn.TUNE_PP8, 1)
        drn.tune(drn.TUNE_PP8_STAGES, 5)
        drn.tune(drn.TUNE_PP8_VARIANT, 1)
        drn.tune(drn.TUNE_PP8_WIDE, 1)
        drn.tune(drn.TUNE_PP8_WIDE_VARIANT, 4)
        drn.tune(drn.TUNE_CONV_RING, ring)
        drn.tune(drn.TUNE_CONV_K2_TILES, k2)
        drn.tune(drn.TUNE_CONV_KSPLIT, ks)
        drn.tune(drn.TUNE_CONV_PP, cpp)
    torch.cuda.synchronize()
    for stages, lst in ys.items():
        for y in lst:
            assert torch.equal(y, tiled), (stages, float((y.float() - tiled.float()).abs().max()))
    got = tiled.float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=2e-2), float((got - ref).abs().max())


@pytest.mark.parametrize("M,N,K,mode,relu", [(2000, 4096, 2048, "drop", True),   # fc7 of R50-C4 at the bench shape: 16 x 32 tiles
                                              (1937, 520, 192, "mask", True),     # ragged M, a ragged column tile, three slabs
                                              (300, 256, 64, "none", False),      # ONE slab, no activation
                                              (77, 136, 4096, "drop", True)])     # fewer rows than a tile, deep K
def test_linear_act_fwd_equals_gemm_then_act(drn, M, N, K, mode, relu):
    """drn_linear_act_fwd (relu_(fc(x)) + dropout as ONE launch of the eight-wave kernel, box_head.py:88-90) == drn_gemm_nt
    (splits = 1, fp32 out) + drn_bias_act_fwd, bit for bit: the bf16 output and its transposed copy (whose rows beyond M stay
    untouched), explicit mask / counter-based dropout with the device-side counter / no dropout, at every ring depth."""
    rs = np.random.RandomState(91)
    dt = torch.bfloat16
    Kp = drn.kpad(K + 24, dt)  # a row pitch beyond K
    A = torch.zeros((M, Kp), dtype=dt, device=DEV)
    A[:, :K] = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32) * 0.3).to(DEV).to(dt)
    W = torch.zeros((N, Kp), dtype=dt, device=DEV)
    W[:, :K] = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32) * (1.0 / math.sqrt(K))).to(DEV).to(dt)
    bias = torch.from_numpy(rs.standard_normal((N,)).astype(np.float32) * 0.1).to(DEV)
    mask, drop_p, seed, seed_dev = None, 0.0, 0, None
    if mode == "mask":
        mask = torch.from_numpy(((rs.rand(M, N) > 0.5) * 2.0).astype(np.float32)).to(DEV)
    elif mode == "drop":
        drop_p, seed = 0.5, 0x1234567
        seed_dev = torch.full((1,), 41, dtype=torch.int64, device=DEV)
    Mp = drn.kpad(M, dt)
    outs = []
    for fused in (0, (3, 0), (4, 0), (5, 0), (5, 2), (5, 1)):  # (ring stages, 256x128 form)
        out = torch.zeros((M, N), dtype=dt, device=DEV)
        outT = torch.full((N, Mp), 7.0, dtype=dt, device=DEV)
        kw = dict(bias=bias, relu=relu, mask=mask, seed=seed, drop_p=drop_p, out=out, outT=outT, seed_dev=seed_dev)
        if fused:
            drn.tune(drn.TUNE_PP8_STAGES, fused[0])
            drn.tune(drn.TUNE_PP8_WIDE, fused[1])
            try:
                assert drn.linear_act_fwd(A, W, M, N, K, **kw)
            finally:
                drn.tune(drn.TUNE_PP8_STAGES, 5)
                drn.tune(drn.TUNE_PP8_WIDE, 1)
        else:
            drn.bias_act_fwd(drn.gemm_nt(A, W, M, N, K), M, N, **kw)
        outs.append((out, outT))
    torch.cuda.synchronize()
    for o, oT in outs[1:]:
        assert torch.equal(o, outs[0][0])
        assert torch.equal(oT, outs[0][1])
    assert torch.equal(outs[0][1][:, :M].t().contiguous(), outs[0][0])
    assert bool((outs[1][1][:, M:] == 7.0).all())
    if seed_dev is not None:
        assert int(seed_dev.item()) == 41  # a dropout launch reads the counter, only the logits pass advances it
    ref = A[:, :K].float().cpu() @ W[:, :K].float().cpu().t() + bias.cpu()
    if relu:
        ref = F.relu(ref)
    if mode == "none":
        assert torch.allclose(outs[1][0].float().cpu(), ref, rtol=2 ** -7, atol=2e-2)
    elif mode == "mask":
        assert torch.allclose(outs[1][0].float().cpu(), ref * mask.cpu(), rtol=2 ** -7, atol=2e-2)
    else:
        keep = (outs[1][0].float().cpu() != 0) | (ref == 0)
        assert abs(float(keep.float().mean()) - (0.5 + 0.5 * float((ref == 0).float().mean()))) < 0.02
        assert torch.allclose(outs[1][0].float().cpu()[keep], (2 * ref)[keep], rtol=2 ** -7, atol=2e-2)
    # outside the kernel's class: refused, not mis-computed
    assert drn.linear_act_fwd(A, W, M, N, K - 32, out=outs[0][0]) is False


@pytest.mark.parametrize("shape", [(375, 500, 480, 640), (375, 500, 1152, 1536), (333, 500, 864, 1297), (500, 375, 240, 180),
                                   (37, 53, 37, 90), (64, 48, 21, 48), (50, 60, 173, 60)])
def test_resize_bilinear_u8_equals_pillow(drn, shape):
    """drn_resize_bilinear_u8 (ResizeTransform [+ HFlipTransform] of the TTA mapper on the device) == Pillow's
    Image.resize(BILINEAR) and the oracle's restatement of it, byte for byte: up- and down-scaling, a direction left
    unchanged, mirrored output."""
    from PIL import Image

    h, w, nh, nw = shape
    rs = np.random.RandomState(h * 7 + nw)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    assert np.array_equal(O.pil_bilinear_resize_u8(img, nh, nw), ref)
    src = torch.from_numpy(img).to(DEV)
    got = drn.resize_bilinear_u8(src, nh, nw)
    assert got.dtype == torch.float32 and tuple(got.shape) == (3, nh, nw)
    assert np.array_equal(got.cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32))
    flipped = drn.resize_bilinear_u8(src, nh, nw, flip=True)
    assert np.array_equal(flipped.cpu().numpy(), np.flip(ref, axis=1).transpose(2, 0, 1).astype(np.float32))


@pytest.mark.parametrize("case", [(1, 99, 151, 256, 1024, True, True),     # dilated-C5 res4 conv3 + shortcut at 800x1216: 59 x 4 tiles, 4 slabs
                                  (1, 99, 151, 512, 2048, True, True),     # res5 conv3 + shortcut: two rounds of tiles
                                  (1, 99, 151, 1024, 2048, False, True),   # res5 projection shortcut (16 slabs)
                                  (2, 131, 97, 64, 520, True, False),      # ONE slab, ragged M, a ragged last column tile, two images
                                  (1, 160, 200, 128, 1000, False, True)])  # Cout % 256 != 0
def test_conv1x1_pp_kernel(drn, case):
    """conv1x1_pp_kernel - a 1x1 / stride-1 bf16 conv of a large map as a GEMM on the 256x256 ping-pong mainloop with the conv
    epilogue (affine, shortcut, ReLU through an fp32 LDS tile, 8-byte row stores) - == the register-staged tiled kernel bit for
    bit (same slab order, k-steps and MFMA per output element) and within bf16 rounding of F.conv2d."""
    n, h, w, cin, cout, has_res, relu = case
    dtype = torch.bfloat16
    x = _rnd((n, cin, h, w), 45)
    wt = _rnd((cout, cin, 1, 1), 46, math.sqrt(2.0 / cin))
    scale, bias = (0.8 + 0.2 * torch.rand(cout)).to(DEV), _rnd((cout,), 47, 0.1).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype)) * scale.cpu().view(1, -1, 1, 1) + bias.cpu().view(1, -1, 1, 1)
    res = None
    if has_res:
        res = _rnd(tuple(ref.shape), 48)
        ref = ref + _q(res, dtype)
        res = res.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    if relu:
        ref = F.relu(ref)
    wp = _pack_w(wt, dtype, drn, cin)
    run = lambda: drn.conv2d_nhwc(xd, wp, cout, 1, 1, 1, 0, 1, scale, bias, res, relu)
    assert drn.tune(drn.TUNE_CONV_PP, 0) == 1
    ring = drn.tune(drn.TUNE_CONV_RING, 0)
    k2 = drn.tune(drn.TUNE_CONV_K2_TILES, 0)
    ks = drn.tune(drn.TUNE_CONV_KSPLIT, 0)
    p8 = drn.tune(drn.TUNE_PP8, 0)
    try:
        tiled = run()  # conv_nhwc_kernel<128x128>
        drn.tune(drn.TUNE_CONV_PP, 2)  # (any layer of >= 2 tiles)
        y = run()
        again = run()
    finally:
        drn.tune(drn.TUNE_CONV_PP, 1)
        drn.tune(drn.TUNE_PP8, p8)
        drn.tune(drn.TUNE_CONV_RING, ring)
        drn.tune(drn.TUNE_CONV_K2_TILES, k2)
        drn.tune(drn.TUNE_CONV_KSPLIT, ks)
    torch.cuda.synchronize()
    assert torch.equal(y, tiled), float((y.float() - tiled.float()).abs().max())
    assert torch.equal(again, y)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=2e-2), float((got - ref).abs().max())


@pytest.mark.parametrize("case", [(1, 203, 181, True), (2, 184, 192, True), (1, 181, 203, False)])
def test_conv3x3_pw_equals_two_convs(drn, case):
    """drn_conv3x3_pw_nhwc (the tail of a res2 bottleneck - 3x3 64 -> 64, ReLU, 1x1 64 -> 256, + shortcut, ReLU - as ONE launch,
    the 3x3's output never in memory) == the two drn_conv2d_nhwc calls, bit for bit; ragged 8 x 32 pixel blocks, two images,
    with and without the shortcut; a map below the kernel's class is refused."""
    n, h, w, has_res = case
    dtype = torch.bfloat16
    x = _rnd((n, 64, h, w), 61)
    w2 = _rnd((64, 64, 3, 3), 62, math.sqrt(2.0 / (64 * 9)))
    w3 = _rnd((256, 64, 1, 1), 63, math.sqrt(2.0 / 64))
    s2, b2 = (0.8 + 0.2 * torch.rand(64)).to(DEV), _rnd((64,), 64, 0.1).to(DEV)
    s3, b3 = (0.8 + 0.2 * torch.rand(256)).to(DEV), _rnd((256,), 65, 0.1).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    res = _rnd((n, h, w, 256), 66).to(DEV).to(dtype) if has_res else None
    wp2, wp3 = _pack_w(w2, dtype, drn, 64), _pack_w(w3, dtype, drn, 64)
    y2 = drn.conv2d_nhwc(xd, wp2, 64, 3, 3, 1, 1, 1, s2, b2, None, True)
    ref = drn.conv2d_nhwc(y2, wp3, 256, 1, 1, 1, 0, 1, s3, b3, res, True)
    got = drn.conv3x3_pw_nhwc(xd, wp2, s2, b2, True, wp3, s3, b3, res, 1.0, True)
    torch.cuda.synchronize()
    assert got is not None
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    assert float(ref.float().abs().max()) > 0
    small = xd[:, :40, :40].contiguous()
    assert drn.conv3x3_pw_nhwc(small, wp2, s2, b2, True, wp3, s3, b3, None, 1.0, True) is None
    # ... with nn.MaxPool2d(2, 2) in the epilogue (odd heights / widths drop their last row / column): the pooled map only
    gotp = drn.conv3x3_pw_nhwc(xd, wp2, s2, b2, True, wp3, s3, b3, res, 1.0, True, pool=True)
    assert gotp is not None and torch.equal(gotp, drn.maxpool2x2_nhwc(ref, 2))
    # ... and without the 1x1 stage (the deep stem's last 3x3 + pool); the 64-channel residual of a basic block
    r64 = _rnd((n, h, w, 64), 67).to(DEV).to(dtype) if has_res else None
    ref64 = drn.maxpool2x2_nhwc(drn.conv2d_nhwc(xd, wp2, 64, 3, 3, 1, 1, 1, s2, b2, r64, True), 2)
    got64 = drn.conv3x3_pw_nhwc(xd, wp2, s2, b2, True, residual=r64, pool=True)
    assert got64 is not None and torch.equal(got64, ref64)
    assert drn.conv3x3_pw_nhwc(xd, wp2, s2, b2, False, pool=True) is None  # a pool without the ReLU in front is refused


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride,hw", [(2, (56, 56)), (1, (28, 28)), (2, (13, 9)), (1, (5, 7))])
def test_maxpool(drn, dtype, stride, hw):
    x = _rnd((2, 16, hw[0], hw[1]), 9)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    y = drn.maxpool2x2_nhwc(xd, stride)
    ref = F.max_pool2d(_q(x, dtype), 2, stride)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref)  # max is exact


@pytest.mark.parametrize("dtype", DTYPES)
def test_preprocess(drn, dtype):
    cfg = O.OracleCfg()
    ims = [torch.randint(0, 256, (3, 40, 52)).float(), torch.randint(0, 256, (3, 33, 60)).float()]
    ref, sizes = O.preprocess_image(ims, cfg)
    cp = 8 if dtype == torch.bfloat16 else 4
    out, sz = drn.preprocess_nhwc([i.to(DEV) for i in ims], cfg.pixel_mean, cfg.pixel_std, dtype, cp)
    assert sz == sizes
    got = out.float().cpu()
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), _q(ref, dtype))
    assert (got[..., 3:] == 0).all()


# ------------------------------------------------------------------------------------------- ROI ops
def _rois(R, n_img, H, W, seed):
    rs = np.random.RandomState(seed)
    x0, y0 = rs.rand(R) * (W - 24), rs.rand(R) * (H - 24)
    r = np.stack([rs.randint(0, n_img, R), x0, y0, x0 + 12 + rs.rand(R) * (W - x0 - 12), y0 + 12 + rs.rand(R) * (H - y0 - 12)], 1)
    r[0, 1:] = [-50, -50, -40, -40]
    r[1, 1:] = [12.0, 20.0, 12.0, 20.0]  # .5 rounding cases at scale 1/8
    return torch.from_numpy(r.astype(np.float32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,P,scale,H,W,R", [(96, 7, 0.125, 23, 29, 80), (6, 3, 0.125, 23, 29, 80),
                                             (130, 7, 0.0625, 23, 29, 80), (128, 7, 0.0625, 14, 14, 83),
                                             (64, 7, 0.125, 40, 37, 80), (64, 7, 0.125, 28, 28, 45),
                                             (16, 7, 0.125, 28, 28, 19), (64, 7, 0.0625, 43, 58, 37),
                                             (24, 7, 0.0625, 75, 100, 21),
                                             (16, 7, 0.0625, 63, 92, 130),   # 64-ROI whole-map kernel, ONE block per CU (143 KB of LDS)
                                             (16, 7, 0.0625, 75, 122, 130)])  # ... the map slice staged in two row bands
def test_roi_pool(drn, dtype, C, P, scale, H, W, R):
    """7x7 pooling of maps that fit in LDS takes the whole-map kernel (ROI groups straddling images, ragged last
    group); C % 64 == 0 otherwise takes the window-staged path; everything else the direct path.  The fused transposed
    output must equal the row-major one whichever kernel produced it."""
    n_img = 2
    feat = _rnd((n_img, C, H, W), 11)
    rois = _rois(R, n_img, W / scale, H / scale, 12)
    obj = torch.rand(R)
    ref, rarg = O.roi_pool_forward(_q(feat, dtype), rois, P, scale)
    ref = _q(ref * (obj + 1).view(-1, 1, 1, 1), dtype).reshape(R, -1)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    out, arg = drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale, want_argmax=True)
    k = C * P * P
    assert torch.equal(out[:, :k].float().cpu(), ref)  # bit-exact: max + one fp32 multiply + rounding
    assert (out[:, k:] == 0).all()
    assert torch.equal(arg.cpu(), rarg.reshape(R, -1))
    out2 = drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale)  # no argmax: fast paths when they apply
    assert torch.equal(out2, out)
    out3 = torch.zeros_like(out)
    out_t = torch.zeros((k, drn.kpad(R, dtype)), dtype=dtype, device=DEV)
    drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale, out=out3, out_t=out_t)
    assert torch.equal(out3, out)
    assert torch.equal(out_t[:, :R], out[:, :k].t())
    assert (out_t[:, R:] == 0).all()


@pytest.mark.parametrize("C,H,W,R,t0", [(1024, 14, 14, 2000, 1003), (64, 14, 14, 200, 40), (128, 50, 76, 130, 127),
                                        (70, 19, 23, 100, 30)])
def test_roi_pool_transposed_tail_hint(drn, C, H, W, R, t0):
    """drn_roi_pool_nhwc_t: with the fc6 dW reading A itself, only the rows of A^T from channel t0 on are needed (the
    columns the dW's tail-balancing launch peels off).  A is unchanged, the rows of A^T from t0 * 49 on equal the full
    transposed copy bit for bit (the 64-ROI kernel skips whole 8-channel chunks below t0, every other kernel writes all)."""
    dtype, P, scale, n_img = torch.bfloat16, 7, 1.0 / 16, 2
    feat = _rnd((n_img, C, H, W), 31)
    rois = _rois(R, n_img, W / scale, H / scale, 32).to(DEV)
    obj = torch.rand(R).to(DEV)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    k = C * P * P
    mk = lambda: (torch.zeros((R, drn.kpad(k, dtype)), dtype=dtype, device=DEV),
                  torch.full((k, drn.kpad(R, dtype)), 7.0, dtype=dtype, device=DEV))
    a_full, t_full = mk()
    drn.roi_pool_nhwc(fd, rois, obj, P, scale, out=a_full, out_t=t_full)
    a_hint, t_hint = mk()
    drn.roi_pool_nhwc(fd, rois, obj, P, scale, out=a_hint, out_t=t_hint, t_first_channel=t0)
    torch.cuda.synchronize()
    assert torch.equal(a_full, a_hint)
    assert torch.equal(t_full[t0 * 49:, :R], t_hint[t0 * 49:, :R])
    assert torch.equal(t_full[:, :R], a_full[:, :k].t())


@pytest.mark.parametrize("C,H,W,R,t0,n_img", [(1024, 14, 14, 2000, 1003, 1), (128, 14, 14, 83, 117, 3), (64, 28, 28, 200, 58, 2),
                                              (128, 50, 76, 300, 120, 2), (16, 63, 92, 130, 15, 2), (24, 40, 37, 65, 22, 4),
                                              (16, 75, 122, 130, 15, 2), (64, 14, 14, 100, 0, 2),
                                              (64, 99, 151, 140, 58, 2),  # the DC5 stride-8 map of an 800x1216 image: 4-channel LDS cells
                                              (128, 63, 92, 300, 120, 3), (64, 43, 58, 150, 58, 3)])  # walking kernel, ragged image runs
def test_roi_pool_lane_kernel_equals_map64(drn, C, H, W, R, t0, n_img):
    """Round 4: the lane-per-bin kernel (a wave per ROI, lane = bin, every channel one 98-byte store run; DRN_TUNE_ROI_LANE)
    writes the bf16 training operand A; the 64-ROI kernel keeps only the A^T tail.  Against the 64-ROI kernel alone and
    against the oracle, bit for bit: ragged image runs inside a block's ROI group, degenerate / clipped boxes, 1 .. 8 channel
    chunks per block, maps from 14x14 to one chunk per CU, and a map too large for it (falls back, same result)."""
    dtype, P, scale = torch.bfloat16, 7, 1.0 / 16
    feat = _rnd((n_img, C, H, W), 41)
    rois = _rois(R, n_img, W / scale, H / scale, 42)
    rois[:, 0] = torch.sort(rois[:, 0]).values if n_img != 3 else rois[:, 0]  # n_img == 3: image index changes every few ROIs
    obj = torch.rand(R)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    k = C * P * P
    res = {}
    # 1: default - maps with one chunk per block take the WALKING kernel (round 5: a window table per block, ROIs from a shared
    # counter, the block walks consecutive chunks); 3: the same with 64 instead of 128 ROIs per block on one-block-per-CU maps and
    # 8 chunks per block forced (the default shortens the walk on grids as small as a test's); 2: the round-4 kernel for those
    # maps; 0: the 64-ROI kernel
    # (lane 1 pools through the chunk-major scratch copy where the walking kernel takes the map - drn_roi_pool_nhwc_ws -, lane 3 without)
    for lane in (3, 2, 1, 0):
        old = drn.tune(19, lane)
        old22 = drn.tune(22, 8) if lane == 3 else None
        drn.ROI_WORKSPACE = lane != 3
        try:
            a = torch.full((R, drn.kpad(k, dtype)), 3.0, dtype=dtype, device=DEV)
            a[:, k:] = 0
            t = torch.full((k, drn.kpad(R, dtype)), 7.0, dtype=dtype, device=DEV)
            drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale, out=a, out_t=t, t_first_channel=t0)
            torch.cuda.synchronize()
            res[lane] = (a, t)
        finally:
            drn.tune(19, old)
            drn.ROI_WORKSPACE = True
            if old22 is not None:
                drn.tune(22, old22)
    ref, _ = O.roi_pool_forward(_q(feat, dtype), rois, P, scale)
    ref = _q(ref * (obj + 1).view(-1, 1, 1, 1), dtype).reshape(R, -1)
    for lane in (3, 2, 1):
        assert torch.equal(res[lane][0], res[0][0]), lane
        assert torch.equal(res[lane][0][:, :k].float().cpu(), ref), lane
        assert torch.equal(res[lane][1][t0 * 49:, :R], res[lane][0][:, t0 * 49: k].t()), lane
        assert torch.equal(res[lane][1][t0 * 49:, :R], res[0][1][t0 * 49:, :R]), lane


def _rois_st(R, n_img, imw, imh, seed):
    """boxes for the sparse-table kernel's cases: every size from a few pixels to the whole image (bins of 1 .. 36 cells: all
    five levels and, beyond 32, the block loop), boxes hanging over every edge (clipped bins: a ROI's smallest bin sets its level,
    the others loop), boxes outside, degenerate boxes, .5 rounding cases"""
    rs = np.random.RandomState(seed)
    w, h = np.exp(rs.rand(R) * np.log(imw / 4.0)) * 4.0, np.exp(rs.rand(R) * np.log(imh / 4.0)) * 4.0
    x0, y0 = rs.rand(R) * (imw - w), rs.rand(R) * (imh - h)
    r = np.stack([rs.randint(0, n_img, R), x0, y0, x0 + w, y0 + h], 1)
    q = R // 8
    r[:q, 1] -= imw * 0.3 * rs.rand(q)       # over the left / top edge
    r[:q, 2] -= imh * 0.3 * rs.rand(q)
    r[q:2 * q, 3] += imw * 0.3 * rs.rand(q)  # over the right / bottom edge
    r[q:2 * q, 4] += imh * 0.3 * rs.rand(q)
    r[2 * q, 1:] = [0, 0, imw, imh]          # the whole image
    r[2 * q + 1, 1:] = [-50, -50, -40, -40]  # outside
    r[2 * q + 2, 1:] = [12.0, 20.0, 12.0, 20.0]
    r[2 * q + 3, 1:] = [imw + 10, 5, imw + 90, 80]
    r[2 * q + 4, 1:] = [-30, imh * 0.5, imw + 30, imh * 0.5 + 3]  # a full-width sliver
    return torch.from_numpy(r.astype(np.float32))


@pytest.mark.parametrize("C,H,W,R,n_img,st,scale", [(64, 99, 151, 400, 2, 1, 0.125),  # the DC5 stride-8 map: 4-channel cells, 15 cells per thread
                                                   (16, 118, 160, 400, 1, 1, 0.125),  # the largest slices that fit (20 cells per thread)
                                                   (8, 70, 255, 400, 3, 1, 0.125),    # 255 columns: bins of 36 cells - three blocks of 16
                                                   (16, 50, 76, 640, 1, 1, 0.0625),   # default rule: 8-channel cells from 3000 cells x 600 ROIs
                                                   (8, 150, 200, 400, 2, 1, 0.125),   # 1200 x 1600 at stride 8: 2-channel cells, 30 cells per thread
                                                   (8, 140, 145, 400, 1, 1, 0.125),   # ... 20 cells per thread
                                                   (8, 81, 101, 120, 10, 2, 0.125),   # ten images, 8-channel cells
                                                   (32, 63, 92, 300, 2, 2, 0.0625),   # forced: 8-channel cells (10 cells per thread)
                                                   (64, 43, 58, 200, 3, 2, 0.0625),   # ... 5 cells per thread
                                                   (16, 30, 40, 64, 1, 2, 0.0625)])
def test_roi_pool_sparse_table_kernel(drn, C, H, W, R, n_img, st, scale):
    """Round 6: RoIPool from a sparse table of block maxima (roi_pool7_st_kernel; DRN_TUNE_ROI_ST) - four table cells per bin
    instead of every cell of the window.  Against the window kernels (knob 0) and the oracle, bit for bit; with and without the
    objectness scale; without the workspace the entry takes the window kernels (same result)."""
    dtype, P = torch.bfloat16, 7
    feat = _rnd((n_img, C, H, W), 51)
    rois = _rois_st(R, n_img, W / scale, H / scale, 52)
    obj = torch.rand(R)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    k = C * P * P
    lib = drn.C.lib()
    res = {}
    for knob in (0, st):
        old = drn.tune(drn.TUNE_ROI_ST, knob)
        try:
            want = lib.drn_roi_pool_workspace_bytes(n_img, H, W, C, P, R, 0, 0, drn.C.dt(dtype), drn.C.dt(dtype))
            if knob:
                assert want >= n_img * H * W * C * 2 + R * 257, "the shape must take the sparse-table kernel"
            a = torch.full((R, drn.kpad(k, dtype)), 3.0, dtype=dtype, device=DEV)
            a[:, k:] = 0
            drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale, out=a)
            b = drn.roi_pool_nhwc(fd, rois.to(DEV), None, P, scale)
            torch.cuda.synchronize()
            res[knob] = (a, b)
        finally:
            drn.tune(drn.TUNE_ROI_ST, old)
    ref, _ = O.roi_pool_forward(_q(feat, dtype), rois, P, scale)
    ref1 = _q(ref * (obj + 1).view(-1, 1, 1, 1), dtype).reshape(R, -1)
    assert torch.equal(res[st][0], res[0][0])
    assert torch.equal(res[st][1], res[0][1])
    assert torch.equal(res[st][0][:, :k].float().cpu(), ref1)
    assert torch.equal(res[st][1][:, :k].float().cpu(), _q(ref, dtype).reshape(R, -1))
    assert (res[st][0][:, k:] == 0).all()
    old = drn.tune(drn.TUNE_ROI_ST, st)
    drn.ROI_WORKSPACE = False
    try:
        c = drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale)
    finally:
        drn.ROI_WORKSPACE = True
        drn.tune(drn.TUNE_ROI_ST, old)
    assert torch.equal(c, res[0][0])


@pytest.mark.parametrize("C,H,W,R", [(2048, 99, 151, 2000),   # the shipped recipe's res5 map of an 800 x 1216 image, every channel
                                     (256, 150, 200, 3000),    # 1200 x 1600 (largest test-time scale): 2-channel cells
                                     (1024, 50, 76, 2000)])    # the C4 map of the same image
def test_roi_pool_sparse_table_full_size(drn, C, H, W, R):
    """The sparse-table kernel at BASELINE's real-image sizes (too large for the CPU oracle in a test): bit-identical to the
    window kernels on the same inputs, and the size-independent properties of a max pool - a constant map pools to the
    constant times the objectness scale wherever a bin is not empty, and pooling is monotone (max(a, b) pooled >= a pooled)."""
    dtype, P, scale = torch.bfloat16, 7, 0.125
    g = torch.Generator().manual_seed(7)
    fd = (torch.randn((1, H, W, C), generator=g) * 0.5).to(DEV).to(dtype)
    rois = _rois_st(R, 1, W / scale, H / scale, 53).to(DEV)
    obj = torch.rand(R, generator=g).to(DEV)
    k = C * P * P
    outs = {}
    for knob in (0, 1):
        old = drn.tune(drn.TUNE_ROI_ST, knob)
        try:
            outs[knob] = drn.roi_pool_nhwc(fd, rois, obj, P, scale)
        finally:
            drn.tune(drn.TUNE_ROI_ST, old)
    assert drn.C.lib().drn_roi_pool_workspace_bytes(1, H, W, C, P, R, 0, 0, drn.C.dt(dtype), drn.C.dt(dtype)) >= H * W * C * 2 + R * 257
    assert torch.equal(outs[1], outs[0])
    const = drn.roi_pool_nhwc(torch.full_like(fd, 1.5), rois, obj, P, scale)[:, :k].float().view(R, C, 49)
    want = (torch.tensor(1.5, device=DEV) * (obj + 1)).to(dtype).float().view(R, 1, 1)
    assert bool(((const == want) | (const == 0)).all())
    assert torch.equal(const[:, 0], const[:, C - 1])  # emptiness is a property of the bin, not of the channel
    hi = drn.roi_pool_nhwc(torch.maximum(fd, fd.flip(1)), rois, None, P, scale)[:, :k].float()
    lo = drn.roi_pool_nhwc(fd, rois, None, P, scale)[:, :k].float()
    assert bool((hi >= lo).all())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("aligned,sr", [(False, 0), (True, 0), (True, 2)])
def test_roi_align(drn, dtype, aligned, sr):
    n_img, C, H, W, P, scale = 2, 70, 19, 23, 7, 0.125
    feat = _rnd((n_img, C, H, W), 13)
    rois = _rois(64, n_img, W / scale, H / scale, 14)
    ref = O.roi_align_forward(_q(feat, dtype), rois, P, scale, sr, aligned).reshape(64, -1)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    out = drn.roi_pool_nhwc(fd, rois.to(DEV), None, P, scale, mode=1, sampling_ratio=sr, aligned=aligned,
                            out_dtype=torch.float32)
    got = out[:, : C * P * P].cpu()
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), float((got - ref).abs().max())


@pytest.mark.parametrize("C,H,W,R,n_img,aligned,sr", [(64, 14, 14, 2000, 1, True, 0),    # the bench map: 8 chunks fit, four per block
                                                      (72, 27, 27, 300, 2, False, 0),   # ragged image runs, C = 9 chunks, ROIAlign (not V2)
                                                      (16, 50, 76, 333, 1, True, 2),    # fixed sampling grid, one chunk per block
                                                      (32, 37, 41, 257, 3, True, 0)])   # three images, odd map
def test_roi_align_lane_kernel(drn, C, H, W, R, n_img, aligned, sr):
    """roi_align7_lane_kernel (bf16 ROIAlign, wave per ROI, lane = bin, the sample's weights computed once per bin for all the
    block's channels) == the generic one-lane-per-channel kernel bit for bit, and == the oracle (pinned to the reference's own
    ROIAlign_cpu.cpp by tests/test_oracle_golden.py) on the bf16-rounded map within one bf16 rounding of the fp32 result;
    objectness scaling, boxes leaving the map, ragged image runs."""
    scale, P, dt = 1.0 / 16, 7, torch.bfloat16
    feat = _rnd((n_img, C, H, W), 23)
    rois = _rois(R, n_img, W / scale, H / scale, 24)
    rois = rois[torch.argsort(rois[:, 0], stable=True)]  # (batches arrive image by image)
    rois[::17, 1] -= 40.0   # boxes that leave the map: samples outside [-1, W] are skipped
    rois[5::29, 4] += 90.0
    obj = torch.rand(R)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    run = lambda: drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale, mode=1, sampling_ratio=sr, aligned=aligned)
    lane = run()
    again = run()
    old = drn.tune(drn.TUNE_ROI_LANE, 0)
    try:
        generic = run()
    finally:
        drn.tune(drn.TUNE_ROI_LANE, old)
    torch.cuda.synchronize()
    k = C * P * P
    assert torch.equal(lane[:, :k], generic[:, :k]), float((lane[:, :k].float() - generic[:, :k].float()).abs().max())
    assert torch.equal(again, lane)
    ref = O.roi_align_forward(_q(feat, dt), rois, P, scale, sr, aligned).reshape(R, -1) * (obj + 1).view(-1, 1)
    got = lane[:, :k].float().cpu()
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=1e-3), float((got - ref).abs().max())


def test_stage_rois_equals_pooler_format(drn):
    """drn_stage_rois (one launch) == convert_boxes_to_pooler_format + the contiguous copies (poolers.py:69-96), bit for bit"""
    g = torch.Generator().manual_seed(3)
    for M in (1, 77, 2000):
        boxes = (torch.rand((M, 4), generator=g) * 500).to(DEV)
        logits = torch.randn(M, generator=g).to(DEV)
        rois, obj, props = drn.stage_rois(boxes, logits, 3.0)
        ref = torch.cat((torch.full((M, 1), 3.0, device=DEV), boxes), dim=1)
        assert torch.equal(rois, ref) and torch.equal(obj, logits) and torch.equal(props, boxes)
        rois2, obj2, props2 = drn.stage_rois(boxes, None)
        assert obj2 is None and torch.equal(rois2[:, 1:], boxes) and (rois2[:, 0] == 0).all() and torch.equal(props2, boxes)


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_cast(drn, dtype):
    x = _rnd((130, 75), 15)
    xd = x.to(DEV)
    t = drn.transpose2d(xd, 130, 75, out_dtype=dtype)
    assert torch.equal(t[:, :130].float().cpu(), _q(x, dtype).t())
    assert (t[:, 130:] == 0).all()
    out = torch.zeros((130, drn.kpad(75, dtype)), dtype=dtype, device=DEV)
    drn.cast2d(xd, 130, 75, out)
    assert torch.equal(out[:, :75].float().cpu(), _q(x, dtype))


# ------------------------------------------------------------------------------------------- epilogues
@pytest.mark.parametrize("dtype", DTYPES)
def test_bias_act_fwd_bwd(drn, dtype):
    M, N, S = 150, 100, 3
    parts = _rnd((S, M, N), 16)
    bias = _rnd((N,), 17)
    mask = (torch.rand(M, N) > 0.5).float() * 2
    out = torch.zeros((M, drn.kpad(N, dtype)), dtype=dtype, device=DEV)
    outT = torch.zeros((N, drn.kpad(M, dtype)), dtype=dtype, device=DEV)
    drn.bias_act_fwd(parts.to(DEV), M, N, bias.to(DEV), True, mask.to(DEV), out=out, outT=outT)
    pre = ((parts[0] + parts[1]) + parts[2]) + bias
    ref = F.relu(pre) * mask
    assert torch.equal(out[:, :N].float().cpu(), _q(ref, dtype))
    assert torch.equal(outT[:, :M].float().cpu(), _q(ref, dtype).t())
    g = _rnd((M, N), 18)
    cs = _rnd((N,), 19, 0.1) * 0 + 0.25
    dpre = torch.zeros_like(out)
    dpreT = torch.zeros_like(outT)
    colsum = torch.zeros((N,), device=DEV)
    drn.bias_act_bwd(g.to(DEV), M, N, saved=out, mask=mask.to(DEV), colscale=cs.to(DEV), dpre=dpre, dpreT=dpreT,
                     colsum=colsum)
    refg = g * cs * mask * (ref > 0).float()
    assert torch.equal(dpre[:, :N].float().cpu(), _q(refg, dtype))
    assert torch.equal(dpreT[:, :M].float().cpu(), _q(refg, dtype).t())
    assert torch.allclose(colsum.cpu(), refg.sum(0), rtol=1e-4, atol=1e-4)
    # counter-based dropout: same mask in fwd and (implicitly) bwd, keep-rate ~ 1-p
    o2 = torch.zeros((M, drn.kpad(N, torch.float32)), dtype=torch.float32, device=DEV)
    drn.bias_act_fwd(torch.ones((1, M, N), device=DEV), M, N, None, True, None, seed=123, drop_p=0.5, out=o2)
    vals = o2[:, :N].cpu()
    assert set(vals.unique().tolist()) <= {0.0, 2.0}
    assert abs(float((vals > 0).float().mean()) - 0.5) < 0.03


@pytest.mark.parametrize("M,N,K,mode", [(2000, 4096, 128, "drop"), (77, 128, 64, "mask"), (300, 256, 256, "none"),
                                         (1000, 192, 192, "drop")])
def test_gemm_nt_act_bwd_equals_gemm_then_act_bwd(drn, M, N, K, mode):
    """drn_gemm_nt_act_bwd (the predictor's dX with fc7's activation backward behind it, one launch, the fp32 product never
    in memory) == drn_gemm_nt (fp32 out) + drn_bias_act_bwd, bit for bit: dpre, its transposed copy, the column sums"""
    rs = np.random.RandomState(57)
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32) * 0.3).to(DEV).to(torch.bfloat16)
    A[:, K - 20:] = 0  # the K padding of the real call
    B = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32) * 0.3).to(DEV).to(torch.bfloat16)
    saved = torch.from_numpy(np.maximum(rs.standard_normal((M, N)), 0).astype(np.float32)).to(DEV).to(torch.bfloat16)
    mask = None
    drop_p = 0.0
    if mode == "mask":
        mask = torch.from_numpy(((rs.rand(M, N) > 0.5) * 2.0).astype(np.float32)).to(DEV)
    elif mode == "drop":
        drop_p = 0.5
    Mp = (M + 63) // 64 * 64
    res = []
    for fused in (False, True):
        dpre = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        dpreT = torch.zeros((N, Mp), dtype=torch.bfloat16, device=DEV)
        colsum = torch.full((N,), 0.25, device=DEV)
        kw = dict(saved=saved if mode != "none" else None, mask=mask, drop_p=drop_p, dpre=dpre, dpreT=dpreT, colsum=colsum,
                  accumulate_colsum=True)
        if fused:
            assert drn.gemm_nt_act_bwd(A, B, M, N, K, **kw)
        else:
            prod = drn.gemm_nt(A, B, M, N, K)
            drn.bias_act_bwd(prod, M, N, **kw)
        res.append((dpre, dpreT, colsum))
    for a, b, name in zip(res[0], res[1], ("dpre", "dpreT", "colsum")):
        assert torch.equal(a, b), name
    assert float(res[1][0].float().abs().max()) > 0
    # outside the kernel's class: refused, not mis-computed
    assert drn.gemm_nt_act_bwd(A, B, M, N, 320, saved=saved, dpre=res[1][0]) is False


# ------------------------------------------------------------------------------------------- MIL head
def _head_inputs(M_per, K, seed):
    rs = np.random.RandomState(seed)
    M = sum(M_per)
    ld = 2 * K + 3 * (K + 1) + 5
    logits = torch.from_numpy(rs.standard_normal((M, ld)).astype(np.float32) * 2)
    off = torch.tensor([0] + list(np.cumsum(M_per)), dtype=torch.int32)
    return logits, off, ld


@pytest.mark.parametrize("K,M_per", [(20, [300]), (5, [40, 35, 61]), (80, [500, 123])])
def test_wsddn_fwd_bwd(drn, K, M_per):
    logits, off, ld = _head_inputs(M_per, K, 21)
    n_img = len(M_per)
    oh = torch.zeros(n_img, K)
    for i in range(n_img):
        oh[i, (3 * i + 1) % K] = 1
        oh[i, (7 * i + 2) % K] = 1
    lg = logits.clone().requires_grad_(True)
    sc = torch.cat([F.softmax(x[:, :K], 1) * F.softmax(x[:, K:2 * K], 0) for x in lg.split(M_per)], 0)
    loss = O.wsddn_loss(sc, M_per, oh, True)
    loss.backward()
    dl = torch.zeros((sum(M_per), ld), device=DEV)
    scores, img_scores, lp = drn.wsddn_fwd_bwd(logits.to(DEV), 0, K, K, off.to(DEV), n_img, oh.to(DEV), dlogits=dl,
                                               max_rows=max(M_per))
    assert torch.allclose(scores.cpu(), sc.detach(), rtol=1e-5, atol=1e-9)
    assert torch.allclose(img_scores.cpu(), O.predict_probs_img(sc.detach(), M_per), rtol=1e-5)
    assert abs(float(lp.sum()) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach()))
    assert torch.allclose(dl.cpu()[:, :2 * K], lg.grad[:, :2 * K], rtol=1e-4, atol=1e-8)


def _boxes(R, seed, W=200, H=150):
    rs = np.random.RandomState(seed)
    x0, y0 = rs.rand(R) * (W - 30), rs.rand(R) * (H - 30)
    return torch.from_numpy(np.stack([x0, y0, x0 + 10 + rs.rand(R) * (W - x0 - 10), y0 + 10 + rs.rand(R) * (H - y0 - 10)],
                                     1).astype(np.float32))


@pytest.mark.parametrize("K,M_per,with_bg", [(20, [700], False), (6, [90, 77], True), (20, [2000], True)])
def test_oicr_targets(drn, K, M_per, with_bg):
    M, n_img = sum(M_per), len(M_per)
    rs = np.random.RandomState(31)
    ncol = K + 1 if with_bg else K
    prev = torch.from_numpy(rs.rand(M, ncol).astype(np.float32))
    props = _boxes(M, 32)
    prev[5] = prev[3]
    props[5] = props[3]  # exact tie -> first index
    if with_bg:
        pb = O.apply_deltas(torch.zeros(M, 4 * K), props)
    else:
        pb = props
    gts = [torch.unique(torch.from_numpy(rs.randint(0, K, 3))).long() for _ in range(n_img)]
    img_scores = torch.from_numpy(rs.rand(n_img, K).astype(np.float32))
    cfg = O.OracleCfg(num_classes=K)
    pgt = O.get_pgt(list(pb.split(M_per)), list(prev.split(M_per)), gts, img_scores, K)
    rl, rw, rm, rb = [], [], [], []
    for (b, c, s, w, idx), p in zip(pgt, props.split(M_per)):
        gc, matched, gb = O.label_proposals(p, b, c, K, cfg)
        rl.append(gc); rm.append(matched); rw.append(w[matched]); rb.append(gb)
    gmax = 4
    gcl = torch.zeros((n_img, gmax), dtype=torch.int32)
    gcn = torch.zeros((n_img,), dtype=torch.int32)
    for i, g in enumerate(gts):
        gcl[i, : len(g)] = g.int()
        gcn[i] = len(g)
    off = torch.tensor([0] + list(np.cumsum(M_per)), dtype=torch.int32)
    out = drn.oicr_targets(prev.to(DEV), pb.to(DEV), props.to(DEV), off.to(DEV), n_img, gcl.to(DEV), gcn.to(DEV),
                           img_scores.to(DEV), K)
    assert torch.equal(out["labels"].cpu().long(), torch.cat(rl))
    assert torch.equal(out["matched"].cpu().long(), torch.cat(rm))
    assert torch.equal(out["weights"].cpu(), torch.cat(rw))
    assert torch.equal(out["gt_boxes"].cpu(), torch.cat(rb))
    for i, (b, c, s, w, idx) in enumerate(pgt):
        assert out["pgt_idx"][i, : len(idx)].cpu().tolist() == idx.tolist()


@pytest.mark.parametrize("K,M_per,nh", [(20, [2000], 3), (6, [90, 77], 2), (80, [1500, 500], 4)])
def test_oicr_refine_chain_equals_per_head_sequence(drn, K, M_per, nh):
    """drn_oicr_refine_chain (all branches in four launches) == targets -> softmax-CE per branch, bit for bit: labels,
    matches, weights, mined boxes, probabilities, losses and the gradient of the logits"""
    M, n_img = sum(M_per), len(M_per)
    rs = np.random.RandomState(77)
    C_ = K + 1
    ld = 2 * K + nh * C_ + 5
    col0s = [2 * K + k * C_ for k in range(nh)]
    logits = torch.from_numpy(rs.standard_normal((M, ld)).astype(np.float32) * 3).to(DEV)
    scores0 = torch.from_numpy(rs.rand(M, K).astype(np.float32)).to(DEV)
    props = _boxes(M, 33).to(DEV)
    img_scores = torch.from_numpy(rs.rand(n_img, K).astype(np.float32)).to(DEV)
    gmax = 4
    gcl = torch.zeros((n_img, gmax), dtype=torch.int32)
    gcn = torch.zeros((n_img,), dtype=torch.int32)
    for i in range(n_img):
        g = torch.unique(torch.from_numpy(rs.randint(0, K, 3)))
        gcl[i, : len(g)] = g.int()
        gcn[i] = len(g)
    gcl, gcn = gcl.to(DEV), gcn.to(DEV)
    off = torch.tensor([0] + list(np.cumsum(M_per)), dtype=torch.int32, device=DEV)
    dl_a = torch.zeros((M, ld), device=DEV)
    dl_b = torch.zeros((M, ld), device=DEV)
    chain = drn.oicr_refine_chain(logits, col0s, K, scores0, props, off, n_img, gcl, gcn, img_scores, dlogits=dl_a)
    prev, zero = scores0, False
    for k in range(nh):
        tg = drn.oicr_targets(prev, props, props, off, n_img, gcl, gcn, img_scores, K, zero_delta_decode=zero)
        probs, loss = drn.softmax_ce(logits, col0s[k], C_, tg["labels"], tg["weights"], dlogits=dl_b)
        ctg, cprobs, closs = chain[k]
        for key in ("labels", "weights", "matched", "gt_boxes"):
            assert torch.equal(ctg[key], tg[key]), (k, key)
        for i in range(n_img):  # only the first G_i mined entries per image are defined
            g = int(gcn[i])
            assert torch.equal(ctg["pgt_idx"][i, :g], tg["pgt_idx"][i, :g]), (k, i)
            assert torch.equal(ctg["pgt_boxes"][i, :g], tg["pgt_boxes"][i, :g]), (k, i)
        assert torch.equal(cprobs, probs) and torch.equal(closs, loss), k
        prev, zero = probs, True
    assert torch.equal(dl_a, dl_b)
    assert float(dl_a.abs().max()) > 0


@pytest.mark.parametrize("K,M_per,nh,splits", [(20, [2000], 3, 8), (6, [90, 77], 2, 3), (80, [1500, 500], 4, 1),
                                               (20, [37, 2000, 5], 3, 4)])
def test_mil_oicr_losses_equals_separate_calls(drn, K, M_per, nh, splits):
    """drn_mil_oicr_losses (six launches) == drn_bias_act_fwd (fp32 logits) + drn_wsddn_fwd_bwd + drn_oicr_refine_chain
    (nine), bit for bit: logits, scores, image scores, losses, targets, probabilities, the gradient of the logits and
    the dropout counter; run twice"""
    M, n_img = sum(M_per), len(M_per)
    rs = np.random.RandomState(91)
    C_ = K + 1
    NH = 2 * K + nh * C_
    ldp = (NH + 7) // 8 * 8
    col0s = [2 * K + k * C_ for k in range(nh)]
    part = torch.from_numpy(rs.standard_normal((splits, M, ldp)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rs.standard_normal(NH).astype(np.float32)).to(DEV)
    props = _boxes(M, 35).to(DEV)
    gmax = 4
    gcl = torch.zeros((n_img, gmax), dtype=torch.int32)
    gcn = torch.zeros((n_img,), dtype=torch.int32)
    oh = torch.zeros((n_img, K))
    for i in range(n_img):
        g = torch.unique(torch.from_numpy(rs.randint(0, K, 3)))
        gcl[i, : len(g)] = g.int()
        gcn[i] = len(g)
        oh[i, g.long()] = 1
    gcl, gcn, oh = gcl.to(DEV), gcn.to(DEV), oh.to(DEV)
    off = torch.tensor([0] + list(np.cumsum(M_per)), dtype=torch.int32, device=DEV)
    # ---- separate calls
    lg_b = torch.zeros((M, ldp), device=DEV)
    ctr_b = torch.full((1,), 5, dtype=torch.int64, device=DEV)
    drn.bias_act_fwd(part, M, NH, bias, False, None, 12345, 0.0, out=lg_b, seed_dev=ctr_b)
    dl_b = torch.zeros((M, ldp), device=DEV)
    sc_b, is_b, lp_b = drn.wsddn_fwd_bwd(lg_b, 0, K, K, off, n_img, oh, dlogits=dl_b, max_rows=max(M_per))
    chain_b = drn.oicr_refine_chain(lg_b, col0s, K, sc_b, props, off, n_img, gcl, gcn, is_b, dlogits=dl_b)
    # ---- the fused tail, twice
    for rep in range(2):
        lg_a = torch.zeros((M, ldp), device=DEV)
        dl_a = torch.zeros((M, ldp), device=DEV)
        ctr_a = torch.full((1,), 5, dtype=torch.int64, device=DEV)
        sc_a, is_a, lp_a, chain_a = drn.mil_oicr_losses(part, bias, lg_a, 0, K, K, off, n_img, oh, col0s, props, gcl, gcn,
                                                        dlogits=dl_a, max_rows=max(M_per), seed_inc=12345, seed_dev=ctr_a)
        assert torch.equal(lg_a[:, :NH], lg_b[:, :NH])
        assert torch.equal(ctr_a, ctr_b) and int(ctr_a) == 5 + 12345
        assert torch.equal(sc_a, sc_b) and torch.equal(is_a, is_b) and torch.equal(lp_a, lp_b)
        for k in range(nh):
            (ta, pa, la), (tb, pb, lb) = chain_a[k], chain_b[k]
            for key in ("labels", "weights", "matched", "gt_boxes"):
                assert torch.equal(ta[key], tb[key]), (k, key)
            for i in range(n_img):
                g = int(gcn[i])
                assert torch.equal(ta["pgt_idx"][i, :g], tb["pgt_idx"][i, :g]), (k, i)
                assert torch.equal(ta["pgt_boxes"][i, :g], tb["pgt_boxes"][i, :g]), (k, i)
            assert torch.equal(pa, pb), k
            assert torch.equal(la, lb), (k, float(la), float(lb))
        assert torch.equal(dl_a, dl_b)
    assert float(dl_a.abs().max()) > 0


@pytest.mark.parametrize("K,M", [(20, 2000), (5, 77), (80, 4000)])
def test_softmax_ce(drn, K, M):
    rs = np.random.RandomState(41)
    ld, col0 = K + 12, 7
    logits = torch.from_numpy(rs.standard_normal((M, ld)).astype(np.float32) * 3)
    labels = torch.from_numpy(rs.randint(0, K + 1, M)).long()
    labels[::17] = -1
    w = torch.from_numpy(rs.rand(M).astype(np.float32))
    w[::5] = 0
    lg = logits.clone().requires_grad_(True)
    loss = O.oicr_cls_loss(lg[:, col0: col0 + K + 1], labels, w)
    loss.backward()
    dl = torch.zeros((M, ld), device=DEV)
    probs, l = drn.softmax_ce(logits.to(DEV), col0, K + 1, labels.int().to(DEV), w.to(DEV), dlogits=dl)
    assert torch.allclose(probs.cpu(), F.softmax(logits[:, col0: col0 + K + 1], -1), rtol=1e-5, atol=1e-9)
    assert abs(float(l) - float(loss.detach())) <= 2e-5 * abs(float(loss.detach()))
    assert torch.allclose(dl.cpu(), lg.grad, rtol=1e-4, atol=1e-9)
    p2 = drn.mean_softmax(logits.to(DEV), [0, 3, 7], K + 1).cpu()
    ref = sum(F.softmax(logits[:, c: c + K + 1], -1) for c in (0, 3, 7)) / 3
    assert torch.allclose(p2, ref, rtol=1e-5, atol=1e-9)
    # (round 6) the wave-per-row kernel sums the denominator in class order like the thread-per-row one: the same bits
    p2b = drn.mean_softmax(logits.to(DEV), [0, 3, 7], K + 1, bg_first=True).cpu()
    old = drn.tune(drn.TUNE_MSM_WAVE, 0)
    try:
        p3 = drn.mean_softmax(logits.to(DEV), [0, 3, 7], K + 1).cpu()
        p3b = drn.mean_softmax(logits.to(DEV), [0, 3, 7], K + 1, bg_first=True).cpu()
    finally:
        drn.tune(drn.TUNE_MSM_WAVE, old)
    assert torch.equal(p2, p3) and torch.equal(p2b, p3b)
    assert torch.equal(p2b[:, :-1], p2[:, 1:]) and torch.equal(p2b[:, -1], p2[:, 0])


def test_apply_deltas_bit_exact(drn):
    K = 6
    props = _boxes(500, 51)
    d = torch.from_numpy(np.random.RandomState(52).standard_normal((500, 4 * K + 3)).astype(np.float32))
    d[0, 3 + 2] = 60.0  # hits the scale clamp
    ref = O.apply_deltas(d[:, 3: 3 + 4 * K].contiguous(), props)
    out = drn.apply_deltas(d.to(DEV), props.to(DEV), K, col0=3)  # deltas start at column 3 of a wider buffer
    assert torch.allclose(out.cpu(), ref, rtol=2e-6, atol=1e-4)  # expf may differ by <= 2 ulp from the CPU's
    z = drn.apply_deltas(None, props.to(DEV), K)
    assert torch.equal(z.cpu(), O.apply_deltas(torch.zeros(500, 4 * K), props))  # zero-delta path: bit-exact


def test_sgd_step(drn):
    cfg = O.OracleCfg()
    rs = np.random.RandomState(61)
    sizes = [1000, 37, 5000, 64]
    names = ["a.weight", "a.bias", "b.weight", "b.bias"]
    p = {n: torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for n, s in zip(names, sizes)}
    flat = torch.cat([p[n] for n in names]).to(DEV)
    mom = torch.zeros_like(flat)
    segs = np.zeros(len(names), dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    o = 0
    for i, (n, s) in enumerate(zip(names, sizes)):
        b = n.endswith("bias")
        segs[i] = (o, s, cfg.base_lr * (cfg.bias_lr_factor if b else 1), cfg.weight_decay_bias if b else cfg.weight_decay)
        o += s
    segs_dev = torch.from_numpy(segs.view(np.uint8)).to(DEV)
    opt = O.SGDState(cfg)
    shadow = torch.zeros(flat.shape, dtype=torch.bfloat16, device=DEV)
    for step in range(3):
        g = {n: torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for n, s in zip(names, sizes)}
        opt.step(p, g)
        drn.sgd_step(flat, mom, torch.cat([g[n] for n in names]).to(DEV), segs_dev, len(names), cfg.momentum, step == 0,
                     shadow=shadow)
        ref = torch.cat([p[n] for n in names])
        assert torch.equal(flat.cpu(), ref)  # same fp32 op order => bit-exact
        assert torch.equal(shadow.float().cpu(), ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("tile", [64, 128, 256])
def test_gemm_nt_bf16_output(drn, tile):
    """c_dtype = bf16 (gradient buckets exchanged in bf16): same accumulators, rounded once (RNE) on the way out"""
    rs = np.random.RandomState(8)
    M, N, K = 300, 520, 256
    A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    B = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    old = drn.gemm_set_tile(tile)
    try:
        ref = drn.gemm_nt(A, B, M, N, K)[0]
        out = torch.full((1, M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        drn.gemm_nt(A, B, M, N, K, out=out)
    finally:
        drn.gemm_set_tile(old)
    assert torch.equal(out[0], ref.to(torch.bfloat16))


def test_sgd_step_bf16_bucket(drn):
    """bf16 gradient bucket covering one tensor of the arena (grad_off) == fp32 step on the same (rounded) gradient"""
    rs = np.random.RandomState(9)
    n0, n1 = 1000, 4096 + 3
    w = torch.from_numpy(rs.standard_normal(n0 + n1).astype(np.float32)).to(DEV)
    seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    seg[0] = (n0, n1, 0.01, 5e-4)
    seg_dev = torch.from_numpy(seg.view(np.uint8)).to(DEV)
    wa, wb = w.clone(), w.clone()
    ma, mb = torch.zeros_like(w), torch.zeros_like(w)
    sa, sb = torch.zeros_like(w, dtype=torch.bfloat16), torch.zeros_like(w, dtype=torch.bfloat16)
    for step in range(3):
        g16 = torch.from_numpy(rs.standard_normal(n1).astype(np.float32)).to(DEV).to(torch.bfloat16)
        g32 = torch.zeros_like(w)
        g32[n0:] = g16.float()
        drn.sgd_step(wa, ma, g32, seg_dev, 1, 0.9, step == 0, 0.5, shadow=sa)
        drn.sgd_step(wb, mb, g16, seg_dev, 1, 0.9, step == 0, 0.5, shadow=sb, grad_off=n0)
        assert torch.equal(wa, wb) and torch.equal(ma, mb) and torch.equal(sa, sb)
    assert torch.equal(wa[:n0], w[:n0]) and not torch.equal(wa[n0:], w[n0:])


@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16])
def test_sgd_step_block_equals_flat(drn, gdt):
    """drn_sgd_step_block over a 2-D partition of one tensor (column slabs + a trailing block, the fc6 dW schedule of round
    4) == ONE drn_sgd_step over the tensor, bit for bit: weights, momentum, bf16 shadow; elements outside the tensor are
    untouched; bf16 gradient bucket addressed through grad_off."""
    rs = np.random.RandomState(17)
    n0, rows, ld = 192, 70, 1000  # the tensor starts at arena element n0 (a multiple of 4)
    tot = n0 + rows * ld + 64
    w = torch.from_numpy(rs.standard_normal(tot).astype(np.float32)).to(DEV)
    seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    seg[0] = (n0, rows * ld, 0.01, 5e-4)
    seg_dev = torch.from_numpy(seg.view(np.uint8)).to(DEV)
    wa, wb = w.clone(), w.clone()
    ma, mb = torch.zeros_like(w), torch.zeros_like(w)
    sa, sb = torch.zeros_like(w, dtype=torch.bfloat16), torch.zeros_like(w, dtype=torch.bfloat16)
    blocks = [(0, rows, 768, 1000), (0, rows, 0, 256), (0, 32, 256, 768), (32, rows, 256, 512), (32, rows, 512, 768)]
    for step in range(3):
        g = torch.from_numpy(rs.standard_normal(rows * ld).astype(np.float32)).to(DEV).to(gdt)
        if gdt == torch.float32:  # fp32 gradients live in an arena-shaped buffer (grad_off = 0)
            gfull = torch.zeros_like(w)
            gfull[n0: n0 + rows * ld] = g
            ga, goff = gfull, 0
        else:
            ga, goff = g, n0
        drn.sgd_step(wa, ma, ga, seg_dev, 1, 0.9, step == 0, 0.5, shadow=sa, grad_off=goff)
        for r0, r1, c0, c1 in blocks:
            drn.sgd_step_block(wb, mb, ga, seg_dev, r0, r1 - r0, c0, c1 - c0, ld, 0.9, step == 0, 0.5, shadow=sb, grad_off=goff)
        assert torch.equal(wa, wb) and torch.equal(ma, mb) and torch.equal(sa, sb), step
    assert torch.equal(wb[:n0], w[:n0]) and torch.equal(wb[n0 + rows * ld:], w[n0 + rows * ld:])
    assert not torch.equal(wb[n0: n0 + rows * ld], w[n0: n0 + rows * ld])
    with pytest.raises(Exception):
        drn.sgd_step_block(wb, mb, ga, seg_dev, 0, rows, 2, 8, ld, 0.9, False, shadow=sb, grad_off=goff)  # c0 % 4 != 0


@pytest.mark.parametrize("M,N,wd,K,kb", [(1024, 20480, 5e-4, 2048, 2000), (768, 24576 + 256, 0.0, 2048, 2000),
                                         (2048, 8192 + 512, 1e-4, 2048, 2048), (1024, 20480, 5e-4, 4032, 4000),
                                         (512, 40960 + 256, 1e-4, 2112, 2100),
                                         # fewer than 32 K slabs (fewer than 2048 proposals: real data) - the chunks of the previous
                                         # tile that find no slab follow the mainloop
                                         (1024, 20480, 5e-4, 1408, 1361), (768, 24576 + 256, 0.0, 576, 565),
                                         (1024, 20480, 5e-4, 1984, 1947), (512, 40960 + 256, 1e-4, 128, 100)])
def test_gemm_tn_sgd_equals_unfused_pair(drn, M, N, wd, K, kb):
    """Round 4: drn_gemm_tn_sgd - the fc6 weight gradient (TN form, bf16 bucket) with the optimizer step of every tile applied
    by the same launch, inside the NEXT tile's mainloop (loads / stores interleaved with the LDS-DMA pipeline on counted
    waits) - against drn_gemm_tn into the bucket followed by drn_sgd_step: bucket, weights, momentum and bf16 shadow bit for
    bit over a first step and two momentum steps.  320 / 291 / 272 tiles on 256 resident workgroups: workgroups with one
    tile (first mainloop + drain only) and with two (pipelined update + drain)."""
    rs = np.random.RandomState(23)
    w0 = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).to(DEV) * 0.02
    seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    seg[0] = (0, M * N, 0.01, wd)
    seg_dev = torch.from_numpy(seg.view(np.uint8)).to(DEV)
    wa, ma, sa = w0.clone(), torch.zeros_like(w0), torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    wb, mb, sb = w0.clone(), torch.zeros_like(w0), torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    for step in range(3):
        A = torch.zeros((M, K), dtype=torch.bfloat16, device=DEV)
        A[:, :kb] = torch.from_numpy(rs.standard_normal((M, kb)).astype(np.float32)).to(DEV).to(torch.bfloat16) * 0.1
        Bt = (torch.from_numpy(rs.standard_normal((kb, N)).astype(np.float32)).to(DEV) * 0.1).to(torch.bfloat16)
        ga = torch.zeros((1, M, N), dtype=torch.bfloat16, device=DEV)
        drn.gemm_tn(A, Bt, M, N, K, kb, out=ga)
        drn.sgd_step(wa.view(-1), ma.view(-1), ga.view(-1), seg_dev, 1, 0.9, step == 0, 0.5, shadow=sa.view(-1))
        gb = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
        assert drn.gemm_tn_sgd(A, Bt, M, N, K, kb, gb, wb, mb, sb, seg_dev, 0.9, step == 0, 0.5)
        torch.cuda.synchronize()
        assert torch.equal(ga[0], gb), step
        assert torch.equal(wa, wb) and torch.equal(ma, mb) and torch.equal(sa, sb), step
    assert not torch.equal(wa, w0)
    # outside the shape class nothing is launched
    if K >= 1088:  # (a contraction length that is not whole 64-element slabs)
        assert not drn.gemm_tn_sgd(A[:, :1056].contiguous(), Bt[:1056].contiguous(), M, N, 1056, 1056, gb, wb, mb, sb, seg_dev, 0.9, False)


# ------------------------------------------------------------------------------------------- conv trunk backward
def _relerr(a, b, floor=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,stride,pad,dil,H,W", [(16, 24, 3, 1, 1, 1, 13, 17), (8, 16, 1, 1, 0, 1, 9, 11),
                                                           (16, 16, 3, 1, 2, 2, 14, 14), (3, 8, 3, 2, 1, 1, 20, 22),
                                                           (128, 128, 3, 1, 1, 1, 32, 32), (64, 128, 3, 1, 1, 1, 32, 32),
                                                           (64, 64, 3, 1, 1, 1, 64, 64)])
def test_conv_backward(drn, dtype, cin, cout, k, stride, pad, dil, H, W):
    """Conv2d.backward_nhwc (mask/affine backward -> wgrad GEMM on the transposed im2col -> dgrad conv on flipped
    weights) vs torch autograd of conv2d * scale + bias -> relu on the CPU"""
    from drn_wsod_pytorch_amd import set_precision
    from drn_wsod_pytorch_amd.layers import Conv2d, FrozenBatchNorm2d

    set_precision("bf16" if dtype == torch.bfloat16 else "fp32")
    rs = np.random.RandomState(3)
    n = 2
    conv = Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, dilation=dil, bias=False,
                  norm=FrozenBatchNorm2d(cout)).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(rs.standard_normal(conv.weight.shape).astype(np.float32) * 0.2))
        conv.norm.weight.copy_(torch.from_numpy(rs.rand(cout).astype(np.float32) + 0.5))
        conv.norm.bias.copy_(torch.from_numpy(rs.standard_normal(cout).astype(np.float32) * 0.1))
    conv.weight.grad = torch.zeros_like(conv.weight)
    x = _q(_rnd((n, cin, H, W), 4), dtype)
    cp = conv.cin_pad(dtype)
    xd = torch.zeros((n, H, W, cp), dtype=dtype, device=DEV)
    xd[..., :cin] = x.permute(0, 2, 3, 1).to(DEV).to(dtype)
    y = conv.run_nhwc(xd, relu=True, explicit_backward=True)
    dy = _q(_rnd(tuple(y.permute(0, 3, 1, 2).shape), 5), dtype)
    need_dx = stride == 1
    dx, _ = conv.backward_nhwc(xd, y, dy.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype), True, need_dx, False, False)
    # reference
    xr = x.clone().requires_grad_(True)
    wr = _q(conv.weight.detach().cpu(), dtype).requires_grad_(True)
    scale, bias = conv.norm.folded()
    yr = torch.relu(torch.nn.functional.conv2d(xr, wr, None, stride, pad, dil) * scale.cpu().view(1, -1, 1, 1)
                    + bias.cpu().view(1, -1, 1, 1))
    yr.backward(dy)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
    # the ReLU mask comes from the (rounded) forward output: compare only where the reference is not at the kink
    assert _relerr(conv.weight.grad.cpu().numpy(), wr.grad.numpy()) < tol
    if need_dx:
        assert _relerr(dx[..., :cin].float().cpu().permute(0, 3, 1, 2).numpy(), xr.grad.numpy()) < tol
        assert (dx[..., cin:] == 0).all()
    set_precision("fp32")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride", [1, 2])
def test_maxpool_backward(drn, dtype, stride):
    n, C, H, W = 2, 24, 11, 14
    x = _q(_rnd((n, C, H, W), 6), dtype)
    x[0, :, 2, 3] = x[0, :, 2, 4]  # tie inside a window: the first maximum takes the gradient
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 2, stride)
    dy = _q(_rnd(tuple(yr.shape), 7), dtype)
    yr.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    dx = drn.maxpool2x2_bwd_nhwc(xd, dy.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype), stride)
    ref = _q(xr.grad, dtype) if stride == 2 else xr.grad
    tol = 1e-2 if (dtype == torch.bfloat16 and stride == 1) else 1e-6
    assert _relerr(dx.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy()) <= tol


# ------------------------------------------------------------------------------------------- ROI backward
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,H,W,R", [(70, 19, 23, 60), (128, 14, 14, 200)])
def test_roi_pool_backward(drn, dtype, C, H, W, R):
    """RoIPool backward (scatter to the forward's arg-max) fused with the objectness scaling, vs the oracle's
    sequential scatter-add; atomics change the fp32 summation order only"""
    n_img, P, scale = 2, 7, 0.125
    feat = _rnd((n_img, C, H, W), 21)
    rois = _rois(R, n_img, W / scale, H / scale, 22)
    obj = torch.rand(R)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    out, arg = drn.roi_pool_nhwc(fd, rois.to(DEV), obj.to(DEV), P, scale, want_argmax=True)
    g = _q(_rnd((R, C, P, P), 23), dtype)
    ref = O.roi_pool_backward(g * (obj + 1).view(-1, 1, 1, 1), rois, arg.cpu().reshape(R, C, P, P), (n_img, C, H, W))
    gd = torch.zeros_like(out)
    gd[:, : C * P * P] = g.reshape(R, -1).to(DEV).to(dtype)
    d = drn.roi_pool_backward_nhwc(gd, rois.to(DEV), obj.to(DEV), (n_img, H, W, C), P, scale, argmax=arg)
    got = d.permute(0, 3, 1, 2).cpu()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float(ref.abs().max()) > 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("aligned,sr", [(False, 0), (True, 0), (True, 2)])
def test_roi_align_backward(drn, dtype, aligned, sr):
    """roi_align_backward vs the oracle (pinned to the reference's ROIAlign_cpu.cpp by tests/test_oracle_golden.py)"""
    n_img, C, H, W, P, scale, R = 2, 70, 19, 23, 7, 0.125, 50
    rois = _rois(R, n_img, W / scale, H / scale, 24)
    if aligned:
        rois = rois[2:]  # aligned=True asserts non-negative ROI size on CPU; row 0/1 are degenerate on purpose
    R = rois.shape[0]
    g = _q(_rnd((R, C, P, P), 25), dtype)
    ref = O.roi_align_backward(g, rois, (n_img, C, H, W), P, scale, sr, aligned)
    gd = torch.zeros((R, drn.kpad(C * P * P, dtype)), dtype=dtype, device=DEV)
    gd[:, : C * P * P] = g.reshape(R, -1).to(DEV).to(dtype)
    d = drn.roi_pool_backward_nhwc(gd, rois.to(DEV), None, (n_img, H, W, C), P, scale, mode=1, sampling_ratio=sr,
                                   aligned=aligned)
    got = d.permute(0, 3, 1, 2).cpu()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_roi_layers_autograd(drn):
    """detectron2.layers.ROIAlign / RoIPool as autograd ops (roi_align.py:22-59): input gradients through the HIP
    backward equal the oracle's autograd functions"""
    from drn_wsod_pytorch_amd.layers import ROIAlign, RoIPool

    n_img, C, H, W, P, scale, R = 2, 24, 17, 21, 7, 0.125, 40
    feat = _rnd((n_img, C, H, W), 26)
    rois = _rois(R, n_img, W / scale, H / scale, 27)[2:]
    up = _rnd((rois.shape[0], C, P, P), 28)
    for mod, fn in ((ROIAlign((P, P), scale, 0, True), lambda x: O._RoIAlignFn.apply(x, rois, P, scale, 0, True)),
                    (RoIPool((P, P), scale), lambda x: O._RoIPoolFn.apply(x, rois, P, scale))):
        xr = feat.clone().requires_grad_(True)
        (fn(xr) * up).sum().backward()
        xd = feat.clone().to(DEV).requires_grad_(True)
        out = mod(xd, rois.to(DEV))
        (out * up.to(DEV)).sum().backward()
        assert xd.grad.shape == xr.grad.shape
        assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max()), type(mod).__name__


# ------------------------------------------------------------------------------------------- inference tail
def test_detect_golden_indices(drn):
    d = G.load("ops")
    b, s, c, rows = drn.detect_topk(torch.from_numpy(d["inf_boxes"]).to(DEV), torch.from_numpy(d["inf_scores"]).to(DEV),
                                    (120, 200), 1e-5, 0.3, 100)
    assert np.array_equal(rows.cpu().numpy(), d["inf_out_rows"])
    assert np.array_equal(c.cpu().numpy(), d["inf_out_classes"])
    assert np.array_equal(s.cpu().numpy(), d["inf_out_scores"])
    assert np.array_equal(b.cpu().numpy(), d["inf_out_boxes"])


@pytest.mark.parametrize("R,K,thr", [(2000, 20, 0.3), (500, 5, 0.5), (1, 3, 0.3), (64, 20, 0.8)])
def test_detect_random(drn, R, K, thr):
    rs = np.random.RandomState(71)
    props = _boxes(R, 72)
    boxes = O.apply_deltas(torch.from_numpy(rs.standard_normal((R, 4 * K)).astype(np.float32) * 0.5), props)
    scores = F.softmax(torch.from_numpy(rs.standard_normal((R, K + 1)).astype(np.float32) * 3), 1)
    if R > 10:
        scores[7] = scores[3]
        boxes[7] = boxes[3]
    rb, rs_, rc, rr = O.fast_rcnn_inference_single_image(boxes.clone(), scores.clone(), (150, 200), 1e-5, thr, 100)
    b, s, c, rows = drn.detect_topk(boxes.to(DEV), scores.to(DEV), (150, 200), 1e-5, thr, 100)
    assert torch.equal(rows.cpu(), rr) and torch.equal(c.cpu(), rc)
    assert torch.equal(s.cpu(), rs_) and torch.equal(b.cpu(), rb)


@pytest.mark.parametrize("R,K,levels", [(2500, 20, 32), (700, 20, 4), (2100, 20, 0)])
def test_detect_sort_ties_and_many_tiles(drn, R, K, levels):
    """Round 3: the candidate sort is this library's own LSD radix sort (sort_hist / sort_scan / sort_scatter kernels,
    three stable passes) instead of hipcub's.  The order it must deliver is torch's scores.sort(stable, descending): ties
    keep candidate (row-major) order.  Scores quantised to a few levels make almost every comparison a tie, spread over
    several 4096-element tiles; R*K >= 40000 also takes batched_nms's per-class branch (detectron2/layers/nms.py:19-29).
    Output rows / classes / scores / boxes must equal the oracle's bit for bit."""
    rs = np.random.RandomState(171 + levels)
    props = _boxes(R, 172)
    boxes = O.apply_deltas(torch.from_numpy(rs.standard_normal((R, 4 * K)).astype(np.float32) * 0.5), props)
    scores = F.softmax(torch.from_numpy(rs.standard_normal((R, K + 1)).astype(np.float32) * 2), 1)
    if levels:
        scores = (scores * levels).ceil() / levels  # every value > 0 passes the threshold; `levels` distinct values
    rb, rs_, rc, rr = O.fast_rcnn_inference_single_image(boxes.clone(), scores.clone(), (150, 200), 1e-5, 0.3, 100)
    for _ in range(2):  # twice: the result may not depend on how the workgroups were scheduled
        b, s, c, rows = drn.detect_topk(boxes.to(DEV), scores.to(DEV), (150, 200), 1e-5, 0.3, 100)
        assert torch.equal(rows.cpu(), rr) and torch.equal(c.cpu(), rc)
        assert torch.equal(s.cpu(), rs_) and torch.equal(b.cpu(), rb)


def test_detect_empty_and_all_filtered(drn):
    boxes = torch.zeros((10, 8)) + torch.tensor([0, 0, 5, 5, 0, 0, 5, 5.0])
    scores = torch.zeros((10, 3))
    scores[:, 2] = 1.0  # everything is background => no candidates
    b, s, c, rows = drn.detect_topk(boxes.to(DEV), scores.to(DEV), (20, 20), 1e-5, 0.3, 100)
    assert b.shape == (0, 4) and s.numel() == 0 and rows.numel() == 0
