"""stringdecomposer_amd -- MI355X-native StringDecomposer read x monomer DP hot path."""
__version__ = "0.1.0"
