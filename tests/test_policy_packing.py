"""CPU: the weight layout the fused policy kernel reads (gym_auv_amd/policy.py: pack_linear) is the one include/auv_hip.h
documents for auv_policy_io.params, and the sizes agree with the library's auv_policy_param_floats."""
import torch

from gym_auv_amd import _capi
from gym_auv_amd.policy import HIDDEN, _pad16, pack_linear, pack_linear_bf16


def test_fragment_order_matches_the_header_formula():
    torch.manual_seed(0)
    for out_f, in_f, out_p, in_p in ((256, 186, 256, 192), (2, 64, 16, 64), (64, 128, 64, 128), (1, 64, 16, 64), (256, 6, 256, 32)):
        w = torch.randn(out_f, in_f)
        packed = pack_linear(w, out_p, in_p)
        assert packed.numel() == out_p * in_p
        K = in_p
        for n in range(0, out_p, 5):
            for k in range(0, in_p, 3):
                idx = ((((n // 16) * (K // 32) + k // 32) * 2 + (k % 8) // 4) * 64 + ((k % 32) // 8) * 16 + n % 16) * 4 + k % 4
                want = float(w[n, k]) if (n < out_f and k < in_f) else 0.0
                assert float(packed[idx]) == want, (n, k)


def test_bf16_fragment_order_matches_the_header_formula():
    torch.manual_seed(1)
    for out_f, in_f, out_p, in_p in ((256, 186, 256, 192), (2, 64, 16, 64), (128, 256, 128, 256)):
        w = torch.randn(out_f, in_f)
        packed = pack_linear_bf16(w, out_p, in_p)
        assert packed.dtype == torch.bfloat16 and packed.numel() == out_p * in_p
        K = in_p
        for n in range(0, out_p, 7):
            for k in range(0, in_p, 5):
                idx = (((n // 16) * (K // 32) + k // 32) * 64 + ((k % 32) // 8) * 16 + n % 16) * 8 + k % 8
                want = float(w[n, k].bfloat16()) if (n < out_f and k < in_f) else 0.0
                assert float(packed[idx]) == want, (n, k)


def test_param_buffer_size_matches_the_library():
    lib = _capi.load_library()
    for obs_dim in (6, 70, 186, 262, 774):
        k0 = _pad16(obs_dim)
        per_net = HIDDEN[0] * k0 + HIDDEN[0] + HIDDEN[1] * HIDDEN[0] + HIDDEN[1] + HIDDEN[2] * HIDDEN[1] + HIDDEN[2] + 16 * HIDDEN[2] + 16
        assert int(lib.auv_policy_param_floats(obs_dim)) == 2 * per_net + 4
    assert int(lib.auv_policy_param_floats(0)) == 0
