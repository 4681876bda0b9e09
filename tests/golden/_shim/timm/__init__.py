"""Test-only stand-in for timm==0.9.6 (requirements.txt:1 of the reference).

Only the six symbols the reference model files import (faster_vit.py:13-15) are provided, so
that /root/reference/fastervit/models can be imported unmodified when generating golden vectors.
Never imported by the product package.
"""
__version__ = "0.9.6-shim"
