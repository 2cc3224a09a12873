"""Drop-in for the hot-path subset of the reference's `musetalk` package (museasr.py / musereal.py seams)."""
