"""Board manipulation and move legality, mirroring ``twentyfortyeight/core`` of the reference."""
