"""Test harness: drivers around the library's operators that mirror callers the reference keeps for itself."""
