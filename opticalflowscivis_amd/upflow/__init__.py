"""Drop-in mirror of the reference's UPFlow/ package layout (model/, utils/) for the hot path."""
