"""desco_amd -- MI355X-native hot path of DeSCo (neighborhood counting + gossip propagation).

Host side mirrors the reference's Python API (subgraph_counting.*); compute runs in
libdesco_hip.so (hand-written gfx950 kernels behind the C ABI of include/desco_hip.h).
"""
__version__ = "0.1.0"
