"""packppi_amd: MI355X-native PackPPI-MSC side-chain sampling path."""
__version__ = "0.1.0"
