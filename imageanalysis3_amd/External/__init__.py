"""External — fitting classes (reference: External/Fitting_v4.py)."""
