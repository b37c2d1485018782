"""Placeholder for the reference's legacy Params.py (argparse, imported by nothing).
Configuration lives in Conf.py / conf/*.toml."""
