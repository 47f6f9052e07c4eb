"""MI355X-native STINet graph-convolution hot path."""
__version__ = '0.1.0'
