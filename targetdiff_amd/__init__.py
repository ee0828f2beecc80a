"""targetdiff_amd: MI355X-native implementation of the TargetDiff denoising hot path.

Host side (this package) mirrors the reference's Python call surface for the path
(``ScorePosNet3D.forward`` / ``sample_diffusion``, ``get_refine_net``, ``sample_diffusion_ligand``);
the per-step work (kNN graph, edge gate, 9 equivariant attention layers, type head, posterior update)
is hand-written HIP for gfx950 behind the C ABI declared in ``include/targetdiff_hip.h``.
"""
__version__ = '0.1.0'
