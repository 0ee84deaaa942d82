"""spmm_amd -- MI355X (gfx950) native implementation of the SPMM dual-encoder pretraining step."""
__version__ = "0.1.0"
