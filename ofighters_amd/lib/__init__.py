"""Host-side mirror of the reference's ofighters.lib interface for the hot path
(Couple/Point, Action, Observation, Battleground).  Arithmetic lives in libofx."""
