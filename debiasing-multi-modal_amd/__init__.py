"""MI355X-native CLIP-embedding + debiasing-adapter hot path.

Imported as `dbmm_amd` through the root shim dbmm_amd.py.
"""
__version__ = "0.1.0"
