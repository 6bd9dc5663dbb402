"""coati_amd: an MI355X-native (gfx950) engine for COATI's contrastive + autoregressive training step.

Compute = hand-written HIP kernels in libcoati_hip.so (C ABI, include/coati_hip.h); this package is the host side:
the ctypes binding, the flat-buffer engine, and mirrors of the reference's Python interface for this path
(`coati_amd.models.encoding.clip_e2e.e3gnn_smiles_clip_e2e`, `coati_amd.training.train_coati`, ...).
PyTorch is used for device memory, streams and torch.distributed only."""
__version__ = "0.1.0"
