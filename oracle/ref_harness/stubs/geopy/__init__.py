"""Raising stand-in for `geopy` (not installed). TEST INFRASTRUCTURE ONLY."""
