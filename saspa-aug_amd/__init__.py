"""saspa_aug_amd -- MI355X-native implementation of SaSPA-Aug's augmentation-generation
hot path (reference: run_aug/run_aug.py).  Host code is Python on PyTorch-ROCm (device
memory, streams, torch.distributed); all arithmetic of the path runs in the hand-written
gfx950 kernels of ``libsaspa_hip.so`` (C ABI: include/saspa_hip.h).  There is no CPU or
eager-PyTorch fallback: importing :mod:`saspa_aug_amd._lib` fails loudly when the library
has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""

__version__ = "0.1.0"
