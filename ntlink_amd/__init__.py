"""MI355X-native minimizer sketch + read->contig mapping for the ntLink `pair` stage."""
__version__ = "0.1.0"
