"""i-vit_amd: MI355X-native integer-only ViT inference path."""
