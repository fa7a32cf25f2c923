"""host logic that picks the arithmetic of a layer (no GPU): split-operand kernels where they can run and are the faster
ones, the fp32 MFMA kernels elsewhere; explicit True / False override the module policy but never the kernel's limits."""
from lion_amd import conv_ops, fused_ops


def test_conv_split_policy():
    assert conv_ops.split_supported(64, 64, 32) and conv_ops.split_supported(128, 128, 8)
    assert not conv_ops.split_supported(4, 32, 32)        # Cin % 16 != 0: the first layer stays on the fp32 kernel
    assert not conv_ops.split_supported(64, 48, 32)       # Cout % 32 != 0
    assert not conv_ops.split_supported(64, 64, 12)
    assert conv_ops.use_split(True, 64, 64, 16) and not conv_ops.use_split(False, 64, 64, 16)
    assert not conv_ops.use_split(True, 35, 64, 16)       # forcing cannot make an unsupported shape run
    assert conv_ops.use_split(None, 128, 128, 8) == (conv_ops.SPLIT and conv_ops.SPLIT_MIN_R <= 8)


def test_pointwise_split_policy():
    pol = fused_ops.pw_use_split
    if fused_ops.PW_SPLIT:
        assert pol(None, 32, 192, 128, 2048)              # feature propagation at N = 2048: 65536 columns
        assert pol(None, 32, 320, 256, 512)
        assert not pol(None, 32, 256, 128, 128)           # 4096 columns: a handful of workgroups, latency bound
        assert not pol(None, 32, 35, 32, 32768)           # long and thin: the fp32 kernel already streams it
        assert pol(None, 32, 64, 128, 8192)
        assert not pol(None, 1, 128, 128, 2048)           # B = 1 demo config
    assert pol(True, 1, 128, 128, 16) and not pol(False, 32, 192, 128, 2048)
    assert not pol(True, 2, 4096, 64, 1 << 18)            # Cin * L beyond the 32-bit byte offsets of the kernel
