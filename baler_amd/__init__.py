"""baler_amd -- MI355X-native train / compress / decompress hot path of Baler (drop-in CLI + configs)."""
__version__ = "0.1.0"
