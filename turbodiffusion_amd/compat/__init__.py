"""Drop-in stand-ins for the reference's native extension modules (seam 1 of INTEGRATION.md)."""
