"""Drop-in replacements for the reference's two CPython extension modules
(``cpp_wrappers.cpp_subsampling.grid_subsampling`` and ``cpp_wrappers.cpp_neighbors.radius_neighbors``),
same module paths, function names, argument meaning and error behaviour, running on the MI355X."""
