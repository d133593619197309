"""CPU: the split GEMM's 32-bit addressing guard (gemm_split.hip s6_chunk_rows, exported as dfd_gemm_chunk_rows).

The kernels address activations with 32-bit byte offsets; a launch therefore never spans 2^31 bytes of X and larger
batches are issued as several launches over whole images.  These cases walk the arithmetic at the capacities
dfd_create accepts (max_batch <= 4096) for the B0 layers with the largest rows-per-image x K products."""
import pytest

LIM = (1 << 31) - 1
# (name, rows per image, K): block 0 project, block 2 project, block 1 expand (unfused), head
LAYERS = [("b0.proj", 112 * 112, 32), ("b2.proj", 56 * 56, 144), ("b1.exp", 112 * 112, 16), ("head", 49, 320),
          ("b12.proj", 49, 1152)]


@pytest.fixture(scope="module")
def lib(pkg):
    import os

    if not os.path.exists(pkg._lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return pkg._lib.load()


@pytest.mark.parametrize("name,hw,k", LAYERS)
@pytest.mark.parametrize("batch", [1, 16, 256, 1189, 1190, 1337, 1338, 2675, 4096])
def test_chunks_stay_below_2g_and_cover_the_batch(lib, name, hw, k, batch):
    rows, row_bytes = batch * hw, k * 4
    chunk = lib.dfd_gemm_chunk_rows(rows, row_bytes, hw)
    assert chunk > 0
    assert chunk * row_bytes <= LIM, "one launch would wrap the 32-bit offset"
    if rows * row_bytes <= LIM:
        assert chunk == rows                      # small batches: one launch, as before
    else:
        assert chunk % hw == 0 and chunk < rows   # whole images, so the gate index m / HW stays chunk-relative
        # the largest whole-image chunk that fits
        assert (chunk + hw) * row_bytes > LIM
    launches = -(-rows // chunk)
    assert (launches - 1) * chunk < rows <= launches * chunk


def test_an_image_that_cannot_fit_is_reported(lib):
    assert lib.dfd_gemm_chunk_rows(4 * (1 << 20), 4096, 1 << 20) == -1      # 4 GiB per image
    assert lib.dfd_gemm_chunk_rows(10, 16, 0) == 10                         # rows_per_image <= 0 is treated as 1


def test_tile_count_is_exported(lib):
    assert 16 <= lib.dfd_gemm_tile_count() <= 128
