# round 4, call m: the round's profile set on the final build (kernel trace, FETCH / WRITE PMC, train trace, evaluate loop, WS attack, full bench line), then the SQ counter passes
bash tools/profile_round.sh 2>&1 | tail -10
bash tools/profile_sq.sh r04 2>&1 | tail -16
