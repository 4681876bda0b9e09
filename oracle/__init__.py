"""CPU oracle for the FasterViT HAT hot path -- test infrastructure only (see hat_reference.py)."""
