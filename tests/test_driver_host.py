"""Host logic of the file-writing driver (generate_samples_batch.py:125-187): caption table parsing and the
PCM_24 writer, checked with the standard-library wave reader.  CPU only."""
import wave

import numpy as np

from text_to_sound_synthesis_amd.pipeline import Diffsound, write_wav_pcm24


def test_read_tsv_groups_captions(tmp_path):
    p = tmp_path / "val.csv"
    p.write_text("file_name,caption\na.wav,a dog barks\nb.wav,rain falls\na.wav,\"birds, chirping\"\n")
    caps = Diffsound.read_tsv(str(p))
    assert caps == {"a.wav": ["a dog barks", "birds, chirping"], "b.wav": ["rain falls"]}


def test_pcm24_writer_roundtrip(tmp_path):
    x = np.concatenate((np.sin(np.arange(2205) * 0.05) * 0.7, [1.5, -1.5, 0.0, 1.0, -1.0]))
    path = str(tmp_path / "x.wav")
    write_wav_pcm24(path, x, 22050)
    with wave.open(path, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 3, 22050, x.size)
        raw = np.frombuffer(w.readframes(x.size), dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v)
    assert np.abs(v[:2205] / 8388608.0 - x[:2205]).max() <= 0.5 / 8388608.0 + 1e-12
    assert list(v[-5:]) == [8388607, -8388608, 0, 8388607, -8388608]          # clipped, no wrap-around


def test_dalle_sample_control_flow_with_stub_backends():
    """DALLE.sample's host logic (dalle_spec.py:264-343: keys, filter-ratio loop, token slicing, train() at the end) with
    the HIP-backed pieces replaced by recording stubs -- the GPU test compares the real thing with the reference."""
    import pytest
    import torch
    from text_to_sound_synthesis_amd.modeling.dalle import DALLE

    class _Tr(torch.nn.Module):
        num_timesteps = 10
        calls = []

        def sample(self, **kw):
            self.calls.append(kw)
            return {"content_token": kw["content_token"] + int(10 * kw["filter_ratio"]) if kw["content_token"].numel()
                    else torch.zeros(2, 265, dtype=torch.long), "logits": torch.ones(1)}

    m = object.__new__(DALLE)
    torch.nn.Module.__init__(m)
    m.content_info, m.condition_info = {"key": "image"}, {"key": "text"}
    m.transformer = _Tr()
    tokens = torch.arange(2 * 265).view(2, 265)
    m.prepare_condition = lambda batch, condition=None: {"condition_token": None, "condition_embed_token": "E"}
    m.prepare_content = lambda batch, with_mask=False: {"content_token": tokens, "content_quant": torch.zeros(2, 256, 5, 53)}
    m.decode_to_img = lambda tok, zshape, stage="first": ("img", tok.clone(), tuple(zshape))
    batch = {"image": "MEL", "text": ["a", "b"]}
    out = m.sample(batch, filter_ratio=[0, 0.5], content_ratio=[1], return_logits=True, noise_fn="NF", batch_size=2)
    assert sorted(out) == sorted(["condition", "input_image", "reconstruction_image", "cond1_cont1_fr0_image",
                                  "cond1_cont1_fr0.5_image", "logits"])
    assert out["condition"] == ["a", "b"] and out["input_image"] == "MEL" and m.training
    assert torch.equal(out["reconstruction_image"][1], tokens) and out["reconstruction_image"][2] == (2, 256, 5, 53)
    assert torch.equal(out["cond1_cont1_fr0.5_image"][1], tokens + 5)
    c = _Tr.calls
    assert [k["filter_ratio"] for k in c] == [0, 0.5] and all(k["condition_embed"] == "E" and k["noise_fn"] == "NF" for k in c)
    assert all(k["return_logits"] and not k["return_att_weight"] and k["sample_type"] == "normal" for k in c)
    out = m.sample(batch, return_rec=False, filter_ratio=[0], content_ratio=[0.5])        # sliced tokens only from all-mask
    assert "reconstruction_image" not in out and _Tr.calls[-1]["content_token"].shape == (2, 132)
    with pytest.raises(ValueError):
        m.sample(batch, filter_ratio=[0.5], content_ratio=[0.5])
    with pytest.raises(NotImplementedError):
        m.sample(batch, return_att_weight=True)
    with pytest.raises(NotImplementedError):
        m.sample(batch, sample_type="debug")


def test_padded_row_mode_selection_and_workspace():
    """Host logic of the native denoiser handle (csrc/api.hip rows_per_sample / carve; no device work, runs without a GPU):
    the sampling step pads every sample to 272 rows exactly where the per-sample GEMM program serves the batch -- f16x2
    mode, row padding on, and the grid rule of csrc/gemm_f16x2_ps.hip (ds_gemm_f16x2_ps_choice, mirrored below) picks full
    or half tiles for the N = 1024 GEMMs: B = 64 / 128 (full tiles), 32 (half tiles), not 1 / 8 / 16 -- and the workspace
    query covers it."""
    import ctypes as C

    import torch
    from text_to_sound_synthesis_amd import _lib as L
    lib = L.lib()
    d = L.DenoiserDesc()
    d.n_layer, d.n_embd, d.n_head, d.seq_len, d.cond_len, d.cond_dim = 2, 1024, 16, 265, 77, 512
    d.n_codes, d.n_steps, d.mlp_mult = 256, 100, 4
    dummy = torch.zeros(64)                       # host memory: the handle only stores pointers
    for f in ("tok_emb", "pos_emb", "lnf_g", "lnf_b", "w_logits", "b_logits", "sched"):
        setattr(d, f, dummy.data_ptr())
    n = 2 * L.LP_COUNT
    ptrs = (C.c_void_p * n)(*([dummy.data_ptr()] * n))
    h = C.c_void_p()
    L.check(lib.ds_denoiser_create(C.byref(d), ptrs, C.byref(h)))
    try:
        assert [lib.ds_denoiser_rows_per_sample(h, B) for B in (1, 32, 64)] == [265, 265, 265]       # fp32 mode: never
        scales = (C.c_float * n)(*([1.0] * n))
        L.check(lib.ds_denoiser_set_split_weights(h, 2, ptrs, scales, dummy.data_ptr(), 1.0))        # f16x2 mode
        def choice(B, N=1024):                    # share of the occupied CU-rounds that work; x0.92 for half tiles; floor 0.65
            tf, th = B * (N // 256), 2 * B * (N // 256)
            ef, eh = tf / (-(-tf // 256) * 256), 0.92 * th / (-(-th // 256) * 256)
            return 0 if max(ef, eh) < 0.65 else (1 if ef >= eh else 2)
        got = {B: lib.ds_denoiser_rows_per_sample(h, B) for B in range(1, 131)}
        assert got == {B: 272 if choice(B) else 265 for B in range(1, 131)}, got
        assert [got[B] for B in (1, 8, 16, 20, 24, 32, 48, 64, 100, 128)] == [265, 265, 265, 265, 272, 272, 272, 272, 272, 272]
        assert (choice(32), choice(64), choice(48)) == (2, 1, 1)
        ws = {B: lib.ds_denoiser_workspace_bytes(h, B) for B in (8, 64)}
        assert ws[64] >= 64 * 272 * 1024 * 4 * (1 + 1 + 3 + 1 + 4) and ws[64] > 8 * ws[8] * 0.99
        L.check(lib.ds_denoiser_set_row_padding(h, 0))
        assert lib.ds_denoiser_rows_per_sample(h, 64) == 265
        L.check(lib.ds_denoiser_set_row_padding(h, 1))
        assert lib.ds_denoiser_rows_per_sample(h, 64) == 272
        assert lib.ds_denoiser_set_split_weights(h, 1, ptrs, scales, dummy.data_ptr(), 1.0) != 0     # (the removed bf16 split)
        L.check(lib.ds_denoiser_set_split_weights(h, 0, None, None, None, 1.0))                      # back to fp32: never
        assert lib.ds_denoiser_rows_per_sample(h, 64) == 265
    finally:
        lib.ds_denoiser_destroy(h)


def test_cpu_baseline_reports_what_the_host_gives(tmp_path):
    """the cpu_baseline object names the CPU model, the affinity count and the cgroup CPU quota, and the
    thread sweep stops at what the process may actually use."""
    import os
    import bench
    info = bench.host_cpu_info()
    assert set(info) == {"model", "affinity_cpus", "cgroup_cpu_max", "cgroup_quota_cpus", "numa_nodes_online"}
    assert info["affinity_cpus"] is None or 1 <= info["affinity_cpus"] <= (os.cpu_count() or 1)
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert bench._cgroup_quota(str(v2)) == ("1600000 100000", 16)
    (v2 / "cpu.max").write_text("max 100000\n")
    assert bench._cgroup_quota(str(v2)) == ("max 100000", None)
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench._cgroup_quota(str(v1)) == ("250000 100000", 3)
    assert bench._cgroup_quota(str(tmp_path / "none")) == (None, None)
