"""Test-only no-op for matplotlib: the reference draws a 128x64 inch histogram
(utils/VStrains_Preprocess.py:62-69) that is not part of any parity surface and takes ~9 s."""
