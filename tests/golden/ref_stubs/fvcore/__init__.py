"""Stand-in for the absent `fvcore` dependency — used ONLY by tests/golden/gen_golden.py in the build
container to import the reference. Never shipped, never used by the product."""
__version__ = "0.1.2"
