"""dad-3dheads_amd: MI355X-native FLAME/HeadMesh decode + Sim3DR hot path."""
__version__ = "0.1.0"
