"""TEST INFRASTRUCTURE ONLY: CPU oracles of the `generate` hot path (see oracle/README.md)."""
