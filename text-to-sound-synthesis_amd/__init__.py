"""MI355X-native Diffsound generation path.

Host side mirrors the reference's Python operator interface
(Diffsound / DALLE / DiffusionTransformer / Text2ImageTransformer / VQModel /
Generator); all arithmetic on the path runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in include/diffsound_hip.h.
"""
__version__ = "0.1.0"
