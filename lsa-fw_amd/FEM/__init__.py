"""LSA-FW FEM package, boundary types of the eigen path only (``FEM.utils``)."""
