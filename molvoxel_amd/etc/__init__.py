"""Helpers either side of the hot path: on-disk grid dump."""
