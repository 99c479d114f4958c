"""empty placeholder: imported by the reference, never called on the step() path."""
