"""classes — containers the hot path hands back (reference: classes/preprocess.py:13-251)."""
