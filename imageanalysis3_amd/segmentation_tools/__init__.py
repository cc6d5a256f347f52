"""segmentation_tools — only what the spot-calling path touches (reference: segmentation_tools/cell.py:598-611)."""
