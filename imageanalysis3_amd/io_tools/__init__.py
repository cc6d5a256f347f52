"""io_tools — helpers the spot-calling path imports (reference: io_tools/crop.py, io_tools/load.py)."""
