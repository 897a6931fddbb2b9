"""Runners for the khoice Snakemake DAGs (Snakemake itself is not installed in this image)."""
