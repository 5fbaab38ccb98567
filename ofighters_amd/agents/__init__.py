"""Host-side mirror of ofighters.agents (Agent + scripted bots, bot plug-in protocol)."""
