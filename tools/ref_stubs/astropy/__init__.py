"""Import-time stand-in for astropy (only Starfish.grid_tools imports it; never called)."""
