"""CPU tests of Track X's dispatch rules through rcn_hipx_plan: the training step's own dispatch code run with every launch replaced by a
note (no GPU needed, none touched).  Holds which kernels the BASELINE configurations run on -- the numerics of those kernels are the GPU
tests' business (tests/test_gpu_convnet*.py)."""
import re

import pytest

from bench_convnet import CONFIGS


@pytest.fixture(scope="module")
def plan():
    from mercer_research_amd import build as hipbuild, convnet
    hipbuild.build_x()
    return convnet.plan


def _lines(text):
    return [l.strip() for l in text.splitlines()[1:] if l.strip()]


def test_cifar_fp32_step_runs_on_the_lds_tiled_kernels_with_pooling_fused_both_ways(plan):
    shp, layers, B = CONFIGS["cifar"]
    L = _lines(plan(shp, layers, B, "fp32", "auto"))
    assert L[0].startswith("conv3x3 32x32x3->32 epi 4: k_conv1_fwd_f32<3, 16>")                # first layer, pool fused into its epilogue
    assert L[1].startswith("conv3x3 16x16x32->64 epi 4: k_conv3x3_halo_f32<16, 64>, 1024 items")
    assert L[2].startswith("conv3x3 8x8x64->128 epi 4: k_conv3x3_halo_f32<8, 64>")              # 8-wide maps: two images per block
    assert not any("k_pool_fwd" in l or "k_pool_bwd" in l or "k_relu_bwd" in l for l in L)
    assert sum("k_head_f32" in l for l in L) == 1 and not any("k_softmax_ce" in l for l in L)
    # both input gradients of the pooled convolutions read the pooled-resolution gradient; so do all three weight gradients
    assert sum("pooled-in: k_conv3x3_halo_f32" in l for l in L) == 2
    assert sum("pooled-dZ" in l and l.startswith("wgrad") for l in L) == 3
    assert any(l.startswith("wgrad conv3x3 32x32x3->32 pooled-dZ: k_conv1_wgrad_f32<3, 16>") for l in L)
    assert L[-1].startswith("update: k_reduce_all, 5 layers' slabs in one launch") and sum("k_reduce_all" in l for l in L) == 1


def test_gemm_tiling_keeps_every_layer_on_the_implicit_gemm_kernels(plan):
    shp, layers, B = CONFIGS["cifar"]
    L = _lines(plan(shp, layers, B, "fp32", "gemm"))
    assert not any(re.search(r"halo|k_conv1_|k_head_f32", l) for l in L)
    assert sum("k_pool_fwd" in l for l in L) == 3 and sum("k_pool_bwd" in l for l in L) == 3
    assert any("k_softmax_ce" in l for l in L)


def test_bf16_mode_keeps_the_first_layer_and_the_head_on_fp32_kernels(plan):
    shp, layers, B = CONFIGS["mnist"]
    L = _lines(plan(shp, layers, 4096, "bf16", "auto"))                                           # BASELINE configs[4]
    assert L[0].startswith("bf16 operand copies") and "k_prep_all_bf16" in L[0]
    assert L[1].startswith("conv3x3 28x28x1->32 epi 4: k_conv1_fwd_f32<1, 16>")
    assert any("k_head_f32" in l for l in L) and any(l.startswith("wgrad conv3x3 28x28x1->32 pooled-dZ: k_conv1_wgrad_f32<1, 16>") for l in L)
    assert any("14x14x32->64 epi 4: k_conv3x3_halo_bf16p" in l for l in L)


def test_synth224_bf16_uses_the_resident_weights_form_only_for_the_32_channel_32_wide_layer(plan):
    shp, layers, B = CONFIGS["synth224"]
    L = _lines(plan(shp, layers, B, "bf16", "auto"))
    one_cb = [l for l in L if "k_conv3x3_halo_bf16_1cb<32>" in l]
    assert len(one_cb) == 2 and all("224x224x32->32" in l for l in one_cb)                         # its forward and its input gradient
    assert any(l.startswith("conv3x3 112x112x32->64 epi 2: k_conv3x3_halo_bf16p") for l in L)      # 64-wide tile: the pipelined form
    assert sum(l.startswith("wgrad") and "k_wgrad3x3_halo_bf16" in l for l in L) == 7
    assert any("split-K 64 + k_splitk_epilogue" in l for l in L)                                   # the 50176 -> 10 layer: capped split
    assert L[-1].startswith("update: k_reduce_all, 9 layers' slabs")


def test_plan_reports_shape_errors_like_create():
    import ctypes as C
    from mercer_research_amd import convnet
    lib = convnet.load()
    arr = (convnet.XLayer * 2)()
    arr[0].kind, arr[0].out = convnet.KIND["conv"], 48                                             # not a multiple of 32
    arr[1].kind, arr[1].out = convnet.KIND["dense"], 10
    buf = C.create_string_buffer(4096)
    assert lib.rcn_hipx_plan(8, 8, 3, arr, 2, 4, 0, 1, buf, len(buf)) == -3
    assert b"multiple of 32" in buf.value
    assert lib.rcn_hipx_plan(8, 8, 3, arr, 2, 4, 7, 1, buf, len(buf)) == -1


def test_dense_weight_gradients_are_sized_by_waves(plan):
    """A wave of the generic weight-gradient kernels owns one 32-row k-block x bn columns over a chunk: the dense layer behind a pooled map is
    few waves.  Below two waves per SIMD the column blocks are 32 wide, below one per SIMD (bf16 kernel) the chunks shorten too."""
    shp, layers, _ = CONFIGS["mnist"]
    L = _lines(plan(shp, layers, 4096, "bf16", "auto"))
    assert any(l.startswith("wgrad dense 1x1x3136->128: k_conv_wgrad_bf16<1, 32, 2>, 4 chunks") for l in L)      # 98 x 4 x 4 = 1568 waves, chunks of 1024 kept
    shp, layers, B = CONFIGS["cifar"]
    L = _lines(plan(shp, layers, B, "bf16", "auto"))
    assert any(l.startswith("wgrad dense 1x1x2048->256: k_conv_wgrad_bf16<1, 32, 4>, 2 chunks") for l in L)      # 64 x 8 x 1 = 512 waves -> chunks of 256
    L = _lines(plan(shp, layers, B, "fp32", "auto"))
    assert any(l.startswith("wgrad dense 1x1x2048->256: k_conv_wgrad<1, tile, 32>, 1 chunks") for l in L)
    # a large layer keeps the 64-wide blocks: synth-224's convolutions never reach this kernel, its dense layer has 32 padded columns anyway
    shp, layers, B = CONFIGS["synth224"]
    assert not any("k_conv_wgrad_bf16<3" in l for l in _lines(plan(shp, layers, B, "bf16", "auto")))


def test_bucket_plan_cuts_the_gradient_into_contiguous_slices_in_backward_order(plan):
    """The bucketed gradient step of a data-parallel rank (rcn_hipx_plan_buckets; SURVEY section 5: bucket by layer, overlap with the weight
    gradients of earlier layers): buckets are final in the order the backward pass finishes them, each is ONE contiguous slice of the
    padded flat gradient, together they cover it exactly once, none but a net's only bucket is smaller than asked, and every bucket ends
    in its own reduction launch."""
    for cfg, prec in (("cifar", "fp32"), ("synth224", "bf16"), ("mnist", "bf16")):
        shp, layers, B = CONFIGS[cfg]
        for min_bytes in (0, 1 << 20, 1 << 30):
            text = plan(shp, layers, B, prec, "auto", buckets=min_bytes)
            done = [re.search(r"bucket (\d+) done: grad\[(\d+), \+(\d+)\)", l) for l in text.splitlines() if " done: grad[" in l]
            ks, offs, lens = [int(m.group(1)) for m in done], [int(m.group(2)) for m in done], [int(m.group(3)) for m in done]
            assert ks == list(range(len(ks))) and len(ks) >= 1
            assert offs[-1] == 0 and all(offs[i] == offs[i + 1] + lens[i + 1] for i in range(len(ks) - 1))       # contiguous, from the top down to 0
            if min_bytes == 1 << 30:
                assert len(ks) == 1
            if len(ks) > 1:
                assert all(4 * n >= min_bytes for n in lens)
            assert text.count("k_reduce_all") == len(ks)
            lines = text.splitlines()
            for i, l in enumerate(lines):
                if " done: grad[" in l:
                    assert "k_reduce_all" in lines[i - 1]
    shp, layers, B = CONFIGS["cifar"]
    t = plan(shp, layers, B, "fp32", "auto", buckets=256 << 10)
    assert t.count(" done: grad[") == 2 and "bucket 0: layers 6 .. 6" in t              # the two dense layers (2.1 MB), then the three convolutions (0.37 MB)
    assert plan(shp, layers, B, "fp32", "auto", buckets=1 << 20).count(" done: grad[") == 1     # ... which are less than 1 MiB and join the bucket above them


def test_bf16_storage_plans_name_the_bf16_tensor_kernels_and_refuse_what_they_do_not_cover():
    """RCN_HIPX_BF16_STORED (no GPU needed): the three BASELINE nets are covered -- every convolution on the LDS-tiled bf16 kernels or the
    first layer's own, every pool fused, no k_pool_fwd / k_pool_bwd / k_relu_bwd launch, the dense layer on top of the stage reading and
    writing bf16 maps; a net whose maps are too small for the LDS-tiled weight gradient is refused with the reason."""
    import pytest
    from mercer_research_amd.convnet import ConvNetError, plan
    import bench_convnet as bc
    for name in ("cifar", "mnist", "synth224"):
        in_shape, layers, B = bc.CONFIGS[name]
        text = plan(in_shape, layers, B, precision="bf16_stored")
        assert "stored as bf16" in text
        assert "k_pool_fwd" not in text and "k_pool_bwd" not in text and "k_relu_bwd" not in text, text
        assert "bf16 input map" in text and "bf16 output map" in text
        assert "k_conv_fwd<" not in text and "k_conv3x3_halo_f32" not in text          # no fp32 convolution kernel but the first layer's
    with pytest.raises(ConvNetError, match="bf16 storage"):
        plan((8, 8, 3), (("conv", 32), ("pool",), ("conv", 64), ("pool",), ("dense", 10)), 16, precision="bf16_stored")
