"""SpecVQGAN decode + MelGAN at B=64: time vs decode_chunk.  Run on the GPU box: python tools/decode_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_sound_synthesis_amd import synth
from text_to_sound_synthesis_amd.config import build_model, default_config
from text_to_sound_synthesis_amd.modeling.vocoder import Generator

torch.set_grad_enabled(False)
m = synth.synth_init_(build_model(default_config(n_layer=1)), seed=0).cuda().eval()
voc = synth.synth_init_(Generator(80, 32, 3), seed=0).cuda().eval()
tok = synth.synth_tokens(64, mask_frac=0.0, key="dab").cuda()
for chunk in (16, 32, 64, 8):
    m.content_codec.decode_chunk = chunk
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel = m.decode_to_img(tok, (64, 256, 5, 53))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        w = voc(mel[:, 0], scale=0.5, shift=0.5)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("decode_chunk %2d: decode %.3f s, vocode %.3f s, peak mem %.1f GB" % (chunk, t1 - t0, t2 - t1, torch.cuda.max_memory_allocated() / 1e9), flush=True)
