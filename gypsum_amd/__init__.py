"""gypsum_amd: MI355X-native GPS L1 C/A correlator engine."""
