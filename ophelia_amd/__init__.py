"""ophelia_amd -- MI355X (gfx950) implementation of Ophelia's Text2Mel + SSRN synthesis
hot path behind the reference's own Python API (config/transcript/CLI, encode_text /
synth_codedtext2mel / synth_mel2mag).  All arithmetic runs in hand-written HIP kernels
reached through the C ABI in include/ophelia_hip.h; there is no CPU fallback."""
__version__ = "0.1.0"
