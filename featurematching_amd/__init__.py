"""featurematching_amd - MI355X-native coarse-to-fine feature matching hot path."""
__version__ = "0.1.0"
