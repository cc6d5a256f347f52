"""correction_tools — filter / alignment / translate operators (reference: correction_tools/)."""
