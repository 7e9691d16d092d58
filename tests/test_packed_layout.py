"""Host side of the packed split-plane layout (include/diffsound_hip.h, ds_gemm_desc.a_split): pack / unpack and the
documented address formula.  CPU only."""
import torch


def test_pack_planes_round_trip_and_formula():
    from text_to_sound_synthesis_amd import _lib as L
    R, K = 37, 96
    x2 = torch.arange(2 * R * K, dtype=torch.float32).remainder(2039).half().view(2, R, K)
    xp = L.pack_planes(x2)
    assert xp.numel() == 2 * 48 * K
    assert torch.equal(L.unpack_planes(xp, R, K), x2)
    flat = xp.view(2, -1)
    for r, k in ((0, 0), (5, 9), (17, 40), (36, 95), (12, 31)):       # the address formula of diffsound_hip.h
        off = ((r // 16) * (K // 32) + k // 32) * 512 + (r % 16) * 32 + (((k // 8) % 4) ^ ((r // 4) % 4)) * 8 + k % 8
        assert flat[0, off] == x2[0, r, k] and flat[1, off] == x2[1, r, k]
