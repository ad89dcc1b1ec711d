"""MI355X-native segment-interest training path (drop-in for MMinterest/models)."""
