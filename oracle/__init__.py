"""CPU oracle for the integer-only ViT path -- TEST INFRASTRUCTURE ONLY (see ivit_oracle.c)."""
