"""MI355X-native gradient x attention relevancy-propagation engine (HIP kernels behind the reference's
Python API of hila-chefer/Transformer-MM-Explainability).  See DESIGN.md.

Import name: ``transformer_mm_explainability_amd`` (alias package at the repo root).
"""
__version__ = "0.1.0"
