"""Observation / reward post-transforms (mirror of ``twentyfortyeight/utils``; the matplotlib UI is out of scope)."""
