"""`musetalk` as the reference imports it (museasr.py:8), resolved to the MI355X implementation."""
