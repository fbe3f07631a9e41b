"""CPU: which large products the library's exact-shape selections cover (datr_amd.tuning.tuned_row_counts) and how
datr_amd.gemm.own_big routes the others (the own GEMM family: a training run with multi-scale resizing, or the real C2F
size, sees row counts the selections file has no entry for)."""
import torch

from datr_amd import gemm, tuning


def test_row_counts_of_the_shipped_selections():
    rows = tuning.tuned_row_counts(tuning.RESULTS, gemm.BIG_ROWS)
    # the DA step's 4 x 22 223 encoder tokens, the source-only / teacher half, the layer1 pixel count of 4 images
    assert {88892, 44446, 267200} <= rows
    assert all(r >= gemm.BIG_ROWS for r in rows)
    assert tuning.tuned_row_counts("/nonexistent/file.csv", gemm.BIG_ROWS) is None


def test_row_counts_parse_every_gemm_form(tmp_path):
    f = tmp_path / "sel.csv"
    f.write_text("Validator,PT_VERSION,2.10.0\n"
                 "GemmTunableOp_float_NT,nt_2048_256_50000_ld_2048_256_2048,Gemm_Hipblaslt_1,0.5\n"
                 "GemmAndBiasTunableOp_float_TN,tn_256_70000_2048_ld_2048_2048_256,Default,0.3\n"
                 "GemmStridedBatchedTunableOp_float_TN,tn_384_900_256_B_2_ld_256_512_384,Gemm_Rocblas_-1,0.01\n")
    assert tuning.tuned_row_counts(str(f), 16384) == {50000, 70000}


def test_own_big_routes_untuned_row_counts(monkeypatch):
    class Mat:                                   # what own_big looks at, without a device
        def __init__(self, rows, cols):
            self.shape, self.dtype, self.is_cuda = (rows, cols), torch.float32, True
        def dim(self): return 2
        def stride(self, i): return self.shape[1] if i == 0 else 1
        def data_ptr(self): return 4096
    w = Mat(256, 256)
    monkeypatch.setattr(gemm, "BACKEND", "library")
    monkeypatch.setattr(gemm, "TUNED_ROWS", frozenset({88892}))
    assert not gemm.own_big(Mat(88892, 256), w)          # a row count with selections: the library
    assert gemm.own_big(Mat(87040, 256), w)              # the C2F size: the own family
    assert not gemm.own_big(Mat(4400, 256), w)           # small products stay on the library
    monkeypatch.setattr(gemm, "TUNED_ROWS", None)        # DATR_GEMM_UNTUNED=library / forced library / no file
    assert not gemm.own_big(Mat(87040, 256), w)
    monkeypatch.setattr(gemm, "BACKEND", "own")
    assert gemm.own_big(Mat(88892, 256), w) and gemm.own_big(Mat(87040, 256), w)


def test_own_family_size_limits_route_oversized_products_to_the_library():
    """csrc/gemm_f32.hip returns DATR_EUNSUPPORTED beyond its 32-bit offsets (an operand of 2 GiB, or
    rows x N / rows x ld >= 2^29); `gemm.own_big` must decline those products instead of letting every
    call site raise -- the library ran them before the own family became a default route."""
    from datr_amd import gemm
    ok = gemm.within_kernel_limits
    # the bench shapes: FFN hidden [88892, 2048] = x [88892, 256] @ w1 [2048, 256].T, and its weight gradient
    assert ok([(88892, 256), (2048, 256)], [256, 256])
    assert ok([(88892, 2048), (256, 2048)], [2048, 2048])
    assert ok([(88892, 2048), (88892, 256)], [2048, 256])
    # ~12 images of 800 x 1333 in one encoder call: rows x 2048 reaches 2^29
    assert not ok([(262144, 256), (2048, 256)], [256, 256])
    assert not ok([(262144, 2048), (256, 2048)], [2048, 2048])
    # an operand of 2 GiB (a row slice of a very wide matrix: the stride counts)
    assert not ok([(70000, 256), (256, 256)], [8192, 256])
    # an output-feature count that is not a multiple of 4 (the NT form's N % 4 rule)
    assert not ok([(88892, 256), (9, 256)], [256, 256])
    assert ok([(88892, 256), (12, 256)], [256, 256])
